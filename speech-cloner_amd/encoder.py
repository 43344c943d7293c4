"""Phoneme-posterior encoder with the reference's ``encoder.encoder_spec_phn`` surface.

Mirrors /root/reference/encoder.py:15-388: ``encoder_spec_phn(cfg_d, ds)`` builds
prenet -> CBHG -> dense(n_output) -> softmax/argmax (encoder.py:78-123) from the same flat
config dict (hp/encoder_cfg_d.json keys) and exposes ``restore / save / predict / run /
get_input / get_outputs / eval_acc``.  The TensorFlow session is replaced by eager HIP kernel
launches (modules.py); TF checkpoints are read/written by tf_bundle.py, so ``restore()`` loads
the reference's own ``enc_14_ckpt``.

Differences a caller can see (documented, deliberate):
  * ``inputs``, ``y_pred`` ... are light-weight handles (there is no graph); ``run(var, feed_dict)``
    evaluates the handles this class defines.
  * optional config key ``compute_dtype`` ('float32' default | 'bfloat16').
  * training (``exec_train_step`` / ``exec_calc_metrics`` / ``train``, encoder.py:256-356) runs in
    float32 through training.EncoderTrainer; TensorBoard summaries are not written.
"""
import os
import sys
from collections import namedtuple

import numpy as np

import modules
import tf_bundle
from modules import prenet, CBHG
from aux_func import *      # noqa: F401,F403  (reference does the same: load_cfg_d, save_cfg_d ...)


class Handle:
    """Stand-in for a tf.Tensor / tf.placeholder attribute of the reference's model objects."""

    def __init__(self, name, shape, dtype='float32'):
        self.name, self.shape, self.dtype = name, tuple(shape), dtype

    def __repr__(self):
        return '<Handle {} shape={} dtype={}>'.format(self.name, self.shape, self.dtype)


class encoder_spec_phn:
    def __init__(self, cfg_d={}, ds=None, store=None):
        self.cfg_d = cfg_d
        self.ds = ds

        self.i_global_step = 0
        self.i_epoch = 0
        self.summary_v = []

        self.store = store if store is not None else modules.VariableStore(
            cfg_d.get('compute_dtype', 'float32'),
            device=cfg_d.get('device', 'cuda'))
        self.sess = self.store                       # shared with a decoder, like the TF session
        modules.apply_options(cfg_d.get('kernel_options'))

        self._build_model(input_shape=self.cfg_d['input_shape'],
                          n_output=self.cfg_d['n_output'],
                          embed_size=self.cfg_d['embed_size'],
                          num_conv_banks=self.cfg_d['num_conv_banks'],
                          num_highwaynet_blocks=self.cfg_d['num_highwaynet_blocks'],
                          dropout_rate=self.cfg_d['dropout_rate'],
                          is_training=self.cfg_d['is_training'],
                          scope=self.cfg_d['model_name'],
                          use_Cudnn=self.cfg_d['use_Cudnn'],
                          use_lstm=self.cfg_d['use_lstm'])

        # optimizer state variables exist in every encoder checkpoint (encoder.py:162-169)
        self.opt_state = {'opt/learning_rate': np.float32(cfg_d.get('learning_rate', 1e-3)),
                          'opt/learning_rate_start': np.float32(cfg_d.get('learning_rate', 1e-3)),
                          'opt/learning_rate_decay': np.float32(cfg_d.get('decay', 0.0)),
                          'opt/global_step': np.int32(0), 'opt/epoch': np.int32(0)}
        return None

    # --------------------------------------------------------------------------- model
    def _build_model(self, input_shape=(800, 256), n_output=48, embed_size=None, num_conv_banks=16,
                     num_highwaynet_blocks=4, dropout_rate=0.5, is_training=True, scope="model",
                     use_Cudnn=False, use_lstm=False, reuse=None):
        """encoder.py:78-123: creates the variables (TF names) and the attribute handles."""
        if embed_size is None:
            embed_size = input_shape[-1]
        modules._refuse_cudnn(use_Cudnn, 'encoder_spec_phn')
        self._embed_size = embed_size
        self._scope = scope
        modules.create_stage_variables(self.store, scope, input_shape[-1], embed_size, num_conv_banks,
                                       num_highwaynet_blocks, n_output, use_lstm=use_lstm)
        T = input_shape[0]
        self.inputs = Handle(scope + '/inputs', (None,) + tuple(input_shape))
        self.target = Handle(scope + '/target', (None, T, n_output))
        self.CBHG_out = Handle(scope + '/CBHG_out', (None, T, embed_size))
        self.y_logits = Handle(scope + '/y_logits', (None, T, n_output))
        self.y_pred = Handle(scope + '/y_pred', (None, T, n_output))
        self.y_pred_class = Handle(scope + '/y_pred_class', (None, T), 'int32')
        return None

    def get_input(self):
        return self.inputs

    def get_outputs(self):
        output_nt = namedtuple('output_nt', 'y_pred y_pred_class y_logits CBHG_out')
        return output_nt(self.y_pred, self.y_pred_class, self.y_logits, self.CBHG_out)

    def _to_device(self, x):
        import torch
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(self.store.device, dtype=torch.float32)
        exp = tuple(self.cfg_d['input_shape'])
        if x.dim() != 3 or tuple(x.shape[1:]) != exp:
            raise ValueError(' - ERROR, encoder input must be [N, {}, {}], got {}'.format(exp[0], exp[1], tuple(x.shape)))
        return x.contiguous()

    def forward(self, x, ppg_pad_to=None, ppg_dtype=None):
        """Device forward of encoder.py:101-111.  x: float32 [N, T, C] device tensor.
        Returns dict(CBHG_out, y_logits (f32), y_pred (f32), y_pred_class (int32) and, when
        ``ppg_pad_to`` is given, 'ppg': the posteriors zero-padded to that width in ``ppg_dtype``
        -- the decoder's input layout)."""
        import torch
        c = self.cfg_d
        if c['is_training']:
            raise Exception(' - ERROR, forward() is the inference graph; a model built with is_training=True '
                            'is evaluated through exec_train_step / exec_calc_metrics')
        st = self.store
        if x.dtype != torch.float32 or not x.is_cuda:
            raise ValueError(' - ERROR, encoder.forward wants a float32 device tensor, got {} on {} '
                             '(predict() / run() convert host arrays)'.format(x.dtype, x.device))
        with modules.variable_store(st), modules.variable_scope(self._scope):
            cbhg_out = modules.prenet_CBHG(x, self._embed_size, c['num_conv_banks'], c['num_highwaynet_blocks'],
                                           c['dropout_rate'], False, prenet_scope="prenet", scope="CBHG",
                                           use_Cudnn=c['use_Cudnn'], use_lstm=c['use_lstm'])
            y_logits = modules.dense(cbhg_out, c['n_output'], None, name="y_logits", out_f32=True)
        dual = ppg_pad_to is not None and ppg_dtype == torch.bfloat16
        if dual:                                       # one launch: y_pred and the decoder's padded bf16 input
            y_pred, y_cls, ppg16 = modules.softmax_argmax_dual(y_logits, ppg_pad_to)
        else:
            y_pred, y_cls = modules.softmax_argmax(y_logits)
        out = {'CBHG_out': cbhg_out, 'y_logits': y_logits, 'y_pred': y_pred, 'y_pred_class': y_cls}
        if ppg_pad_to is not None:
            if dual:
                out['ppg'] = ppg16
            elif ppg_pad_to == c['n_output'] and (ppg_dtype is None or ppg_dtype == torch.float32):
                out['ppg'] = y_pred
            else:
                out['ppg'] = modules.softmax_argmax(y_logits, pad_to=ppg_pad_to, out_dtype=ppg_dtype)[0]
        return out

    # --------------------------------------------------------------------------- checkpoints
    def _all_variables(self):
        d = {n: v for n, v in self.store.to_numpy().items() if n.startswith(self._scope + '/')}
        d.update(self.opt_state)
        tr = getattr(self, '_trainer', None)
        if tr is not None:
            d.update(tr.slot_dict())                   # opt/<var>/Adam[_1], opt/beta{1,2}_power
        return d

    def save(self, save_path=None, i_checkpoint=None, verbose=True):
        """encoder.py:223-235: ``{model_path}/{model_name}-{step}`` in TF bundle format."""
        if save_path is None:
            save_path = '{}/{}'.format(self.cfg_d['model_path'], self.cfg_d['model_name'])
        if i_checkpoint is None:
            i_checkpoint = self.i_global_step
        prefix = '{}-{}'.format(save_path, int(i_checkpoint))
        tf_bundle.write_bundle(prefix, self._all_variables())
        tf_bundle.update_checkpoint_state(os.path.dirname(prefix) or '.', os.path.basename(prefix))
        if verbose:
            print(' Saved: "{}"'.format(prefix))
        return None

    def restore(self, save_path=None, i_checkpoint=None):
        """encoder.py:238-253: latest checkpoint of cfg_d['model_path'] (or an explicit step /
        prefix); on failure prints to stderr and exits with status 1, like the reference."""
        if save_path is None:
            if i_checkpoint is None:
                save_path = tf_bundle.latest_checkpoint(self.cfg_d['model_path'])
            else:
                save_path = '{}/{}-{}'.format(self.cfg_d['model_path'], self.cfg_d['model_name'], int(i_checkpoint))
        try:
            names = None if self.cfg_d['is_training'] else \
                set(n for n in self.store.vars if n.startswith(self._scope + '/')) | set(self.opt_state)
            w = tf_bundle.read_bundle(save_path, verify_crc=True, names=names)
            self.store.load_dict({k: v for k, v in w.items() if k in self.store.vars}, strict=False)
            missing = [n for n in self.store.vars if n.startswith(self._scope + '/') and n not in w]
            if missing:
                raise KeyError(missing[:3])
            for k in self.opt_state:
                if k in w:
                    self.opt_state[k] = w[k]
            self.i_global_step = int(self.opt_state['opt/global_step'])
            self.i_epoch = int(self.opt_state['opt/epoch'])
            if self.cfg_d['is_training']:              # Adam slots + step: now, or when the trainer is created
                if getattr(self, '_trainer', None) is not None:
                    self._trainer.resume(w)
                else:
                    self._restored_ckpt = w
            print('Restored: "{}"'.format(save_path))
        except Exception:
            print(' Model not found: {}'.format(save_path), file=sys.stderr)
            sys.exit(1)
        return None

    # --------------------------------------------------------------------------- run API
    def predict(self, x, batch_size=32):
        """encoder.py:359-367: softmax posteriors [N, T, n_output] (numpy float32), evaluated in
        chunks of ``batch_size`` windows."""
        y_pred_v = []
        for i_s in range(0, x.shape[0], batch_size):
            x_batch = self._to_device(x[i_s:min(i_s + batch_size, x.shape[0])])
            y_pred_v.append(self.forward(x_batch)['y_pred'].cpu().numpy())
        return np.concatenate(y_pred_v, axis=0)

    def run(self, var, feed_dict={}):
        """encoder.py:370-371: evaluate one handle or a list of handles given {inputs: x}."""
        single = not isinstance(var, (list, tuple))
        vs = [var] if single else list(var)
        if self.inputs not in feed_dict:
            raise Exception(' - ERROR, run: feed_dict must provide encoder.inputs')
        out = self.forward(self._to_device(feed_dict[self.inputs]))
        res = []
        for v in vs:
            key = v.name.split('/')[-1]
            if key not in out:
                raise Exception(' - ERROR, run: {} cannot be evaluated'.format(v))
            res.append(out[key].float().cpu().numpy() if key != 'y_pred_class' else out[key].cpu().numpy())
        return res[0] if single else res

    def eval_acc(self, ds_sampler, n_batchs=100):
        """encoder.py:374-388."""
        n_c = 0
        n_t = 0
        acc = 0.0
        for i_batch, (mfcc_batch, phn_v_batch) in enumerate(ds_sampler):
            y_pred = self.run(self.y_pred, {self.inputs: mfcc_batch})
            y_dec = np.argmax(y_pred, axis=-1)
            y_true = np.argmax(phn_v_batch, axis=-1)
            n_c += (y_dec == y_true).sum()
            n_t += y_dec.size
            acc = n_c / n_t
            print('acc[{:4d}] = {:5.03f}'.format(int(n_t), acc))
        return acc, n_t

    # --------------------------------------------------------------------------- training
    def _get_trainer(self):
        if not self.cfg_d['is_training']:
            raise Exception('Model is not in training model')
        if getattr(self, '_trainer', None) is None:
            import training
            self._trainer = training.EncoderTrainer(self)
            if getattr(self, '_restored_ckpt', None) is not None:      # resume Adam state like tf.train.Saver
                self._trainer.resume(self._restored_ckpt)
                self._restored_ckpt = None
        return self._trainer

    def _target_to_device(self, target):
        import torch
        if not torch.is_tensor(target):
            target = torch.from_numpy(np.ascontiguousarray(target, dtype=np.float32))
        target = target.to(self.store.device, dtype=torch.float32).contiguous()
        T, n_out = self.cfg_d['input_shape'][0], self.cfg_d['n_output']
        if target.dim() != 3 or tuple(target.shape[1:]) != (T, n_out):
            raise ValueError(' - ERROR, target must be [N, {}, {}], got {}'.format(T, n_out, tuple(target.shape)))
        return target

    def exec_train_step(self, inputs, target):
        """encoder.py:256-270: forward (dropout, batch statistics) + softmax cross-entropy + backward +
        Adam.  Returns (loss, acc, mse, global_step, train_step); train_step is None (a TF op)."""
        import torch
        tr = self._get_trainer()
        out3 = tr.forward_backward(self._to_device(inputs), self._target_to_device(target), backward=True)
        world = torch.distributed.get_world_size() if (torch.distributed.is_available() and
                                                        torch.distributed.is_initialized()) else 1
        global_step = tr.apply_gradients(world)
        loss, acc, mse = (np.float32(v) for v in out3.cpu().numpy())
        self.i_global_step = global_step
        return (loss, acc, mse, np.int32(global_step), None)

    def exec_calc_metrics(self, inputs, target, summary_mode='validation'):
        """encoder.py:274-297 without the TensorBoard writers: (acc, mse, loss).  Like the reference's
        graph, the forward runs in whatever mode the model was built in (training mode keeps dropout
        and batch statistics active and, with updates_collections=None, also moves the averages)."""
        if summary_mode not in ('train', 'validation', 'test'):
            raise Exception(' - ERROR, summary_mode={} not implemented'.format(summary_mode))
        import torch
        x, t = self._to_device(inputs), self._target_to_device(target)
        if self.cfg_d['is_training']:
            out3 = self._get_trainer().forward_backward(x, t, backward=False)
        else:
            import ctypes as C
            import _vc
            y = self.forward(x)['y_logits']
            M, n_out = y.shape[0] * y.shape[1], y.shape[2]
            out3 = torch.empty(3, dtype=torch.float32, device=y.device)
            ws = torch.empty(3 * M, dtype=torch.float32, device=y.device)
            _vc.check(_vc.lib().vc_softmax_ce(C.c_void_p(y.data_ptr()), C.c_void_p(t.data_ptr()), M, n_out, n_out, None, 0,
                                              C.c_void_p(out3.data_ptr()), C.c_void_p(ws.data_ptr()), _vc.current_stream()))
        loss, acc, mse = (np.float32(v) for v in out3.cpu().numpy())
        return acc, mse, loss

    def _lr_decay(self):
        """encoder.py:183: lr = lr_start / (1 + decay * epoch)."""
        o = self.opt_state
        o['opt/learning_rate'] = np.float32(float(o['opt/learning_rate_start']) /
                                            (1.0 + float(o['opt/learning_rate_decay']) * float(o['opt/epoch'])))
        return o['opt/learning_rate']

    def train(self):
        """encoder.py:300-356 (same prints and control flow; needs a dataset object providing
        get_ds_filter / window_sampler, which this package does not ship)."""
        if not self.cfg_d['is_training']:
            raise Exception('Model is not in training model')
        self.cfg_d['n_samples_trn'] = self.ds.get_ds_filter(self.cfg_d['ds_trn_filter_d']).sum()
        self.cfg_d['n_steps_epoch_trn'] = self.cfg_d['n_samples_trn'] // self.cfg_d['batch_size']
        self.sampler_trn = self.ds.window_sampler(batch_size=self.cfg_d['batch_size'], n_epochs=99999999,
                                                  randomize_samples=self.cfg_d['randomize_samples'],
                                                  ds_filter_d=self.cfg_d['ds_trn_filter_d'])
        self.sampler_val = self.ds.window_sampler(batch_size=self.cfg_d['batch_size'], n_epochs=99999999,
                                                  randomize_samples=self.cfg_d['randomize_samples'],
                                                  ds_filter_d=self.cfg_d['ds_val_filter_d'])
        self.iter_val = iter(self.sampler_val)
        print(' Starting Training ...')
        print(' n_samples_trn:    ', self.cfg_d['n_samples_trn'])
        print(' n_steps_epoch_trn:', self.cfg_d['n_steps_epoch_trn'])
        print(' batch_size:       ', self.cfg_d['batch_size'])
        print(' n_epochs:         ', self.cfg_d['n_epochs'])
        input('Press --ENTER--')
        self.i_epoch = int(self.opt_state['opt/epoch'])
        self.lr = self._lr_decay()
        for mfcc_trn, phn_v_trn in self.sampler_trn:
            loss, acc, mse, global_step, train_step = self.exec_train_step(mfcc_trn, phn_v_trn)
            print(' - i_epoch={}   global_step={}   loss_trn={:6.3f}  acc_trn={:6.3f}  mse_trn={:6.3f}'.format(
                self.i_epoch, global_step, loss, acc, mse))
            if (global_step / self.cfg_d['n_steps_epoch_trn']) % self.cfg_d['save_each_n_epochs'] == 0:
                print(' Saving, epoch={} ...'.format(self.i_epoch))
                self.save()
                mfcc_val, phn_v_val = next(self.iter_val)
                acc_val, mse_val, loss_val = self.exec_calc_metrics(mfcc_val, phn_v_val)
                print(' - i_epoch={}   global_step={}   loss_val={:6.3f}  acc_val={:6.3f}   mse_val={:6.3f}'.format(
                    self.i_epoch, int(global_step), loss_val, acc_val, mse_val))
            if global_step % self.cfg_d['n_steps_epoch_trn'] == 0:
                self.opt_state['opt/epoch'] = np.int32(int(self.opt_state['opt/epoch']) + 1)
                self.i_epoch = int(self.opt_state['opt/epoch'])
                self.lr = self._lr_decay()
                if self.i_epoch >= self.cfg_d['n_epochs']:
                    break
        print(' End of Training !!!')
        return None
