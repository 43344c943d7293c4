"""Spectrogram decoder with the reference's ``decoder.decoder_specs`` surface.

Mirrors /root/reference/decoder.py:19-493: ``decoder_specs(cfg_d, ds, encoder)`` grafts two
prenet -> CBHG -> dense stages onto the encoder's posteriors (decoder.py:75-182):
step1 -> y_mel (n_mels), step2 (fed with y_mel, ``use_target_mel_step2`` false) -> y_stft, from
the same config dict (hp/decoder_cfg_d.json keys).  Like the reference it shares the encoder's
"session" (here: the VariableStore) and calls ``encoder.restore()`` in its constructor
(decoder.py:57).  ``predict`` returns namedtuple('predict', 'y_mel y_stft y_phn').
"""
import os
import sys
from collections import namedtuple

import numpy as np

import modules
import tf_bundle
from modules import prenet, CBHG
from encoder import encoder_spec_phn, Handle
from aux_func import *      # noqa: F401,F403


class decoder_specs:
    def __init__(self, cfg_d={}, ds=None, encoder=None):
        self.cfg_d = cfg_d
        self.ds = ds

        self.i_global_step = 0
        self.i_epoch = 0
        self.summary_v = []
        self.spec_summary_v = []

        self.encoder = encoder
        self._create_tf_session()
        modules.apply_options(cfg_d.get('kernel_options'))
        self._build_model(reuse=None)

        self.opt_state = {'dec_opt/learning_rate': np.float32(cfg_d.get('learning_rate', 1e-3)),
                          'dec_opt/learning_rate_start': np.float32(cfg_d.get('learning_rate', 1e-3)),
                          'dec_opt/learning_rate_decay': np.float32(cfg_d.get('decay', 0.0)),
                          'dec_opt/global_step': np.int32(0), 'dec_opt/epoch': np.int32(0)}

        if encoder is not None:
            # decoder.py:57 -- the constructor restores the encoder's weights
            encoder.restore()
            print(' Encoder Restored !!!')
        return None

    def _create_tf_session(self):
        """decoder.py:63-71: share the encoder's session (VariableStore) when there is one."""
        if self.encoder is None:
            self.store = modules.VariableStore(self.cfg_d.get('compute_dtype', 'float32'),
                                               device=self.cfg_d.get('device', 'cuda'))
        else:
            self.store = self.encoder.store
        self.sess = self.store
        return None

    # --------------------------------------------------------------------------- model
    def _step_embed(self, i, prev_E):
        sd = self.cfg_d['steps_v'][i]
        if sd['embed_size'] is None:
            return self.cfg_d['input_shape'][-1] if i == 0 else prev_E
        return sd['embed_size']

    def _build_model(self, reuse=None):
        """decoder.py:75-182: variables (TF names decoder/step{1,2}/...) and attribute handles."""
        c = self.cfg_d
        scope = c['model_name']
        self._scope = scope
        modules._refuse_cudnn(c['use_Cudnn'], 'decoder_specs')
        T, n_in = c['input_shape']
        if self.encoder is None:
            self.inputs = Handle(scope + '/inputs', (None, T, n_in))
        else:
            self.inputs = self.encoder.get_input()
            enc_o = self.encoder.get_outputs()
            assert list(enc_o.y_logits.shape[1:]) == list(c['input_shape']), \
                'ERROR, input_shape no coincide con la dimensión de salida del encoder.'
        self.dec_inputs = Handle(scope + '/dec_inputs', (None, T, n_in))
        self._E = []
        cin = n_in
        prev_E = None
        for i, sd in enumerate(c['steps_v'][:2]):
            E = self._step_embed(i, prev_E)
            self._E.append(E)
            modules.create_stage_variables(self.store, '{}/step{}'.format(scope, i + 1), cin, E,
                                           sd['num_conv_banks'], sd['num_highwaynet_blocks'], sd['n_output'],
                                           use_lstm=c['use_lstm'])
            cin = sd['n_output']
            prev_E = E
        # decoder.py:148-152: teacher-forced stage 2 -- inputs_step2 = f * y_mel + (1 - f) * target_mel with the
        # non-trainable scalar decoder/step2/inputs_step2/f_mel_pred_tf (0.0 at first, raised per epoch, :258-260)
        self._F_NAME = '{}/step2/inputs_step2/f_mel_pred_tf'.format(scope)
        self.f_mel_pred = 0.0
        if c.get('use_target_mel_step2', False):
            self.store.get(self._F_NAME, (), 0.0)
            self.store.non_trainable.add(self._F_NAME)
        n_mel, n_stft = c['steps_v'][0]['n_output'], c['steps_v'][1]['n_output']
        self.y_mel = Handle(scope + '/y_mel', (None, T, n_mel))
        self.target_mel = Handle(scope + '/step1/target', (None, T, n_mel))
        self.y_stft = Handle(scope + '/y_stft', (None, T, n_stft))
        self.target_stft = Handle(scope + '/step2/target', (None, T, n_stft))
        self.mel_loss = Handle(scope + '/mel_loss', ())
        self.stft_loss = Handle(scope + '/stft_loss', ())
        self.loss = Handle(scope + '/loss', ())
        return None

    def _to_device(self, x, width, what):
        import torch
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(self.store.device, dtype=torch.float32)
        T = self.cfg_d['input_shape'][0]
        if x.dim() != 3 or x.shape[1] != T or x.shape[2] != width:
            raise ValueError(' - ERROR, {} must be [N, {}, {}], got {}'.format(what, T, width, tuple(x.shape)))
        return x.contiguous()

    def forward_from_ppg(self, ppg, target_mel=None):
        """Stages step1/step2 on device posteriors ``ppg`` [N, T, pad8(n_in)] (compute dtype, zero
        padding columns).  Returns (y_mel f32, y_stft f32).  With use_target_mel_step2 the second stage is fed
        f_mel_pred * y_mel + (1 - f_mel_pred) * target_mel (decoder.py:152) and ``target_mel`` is required -- like
        the reference's graph, which cannot be evaluated without that placeholder."""
        c = self.cfg_d
        if c['is_training']:
            raise NotImplementedError(' - ERROR, use exec_train_step for training mode')
        blend = c.get('use_target_mel_step2', False)
        if blend and target_mel is None:
            raise Exception(' - ERROR, use_target_mel_step2: the second stage needs target_mel (feed decoder.target_mel)')
        st = self.store
        x = ppg
        cin = c['input_shape'][-1]
        ys = []
        with modules.variable_store(st), modules.variable_scope(self._scope):
            for i, sd in enumerate(c['steps_v'][:2]):
                with modules.variable_scope('step{}'.format(i + 1)):
                    pre = prenet(inputs=x, num_units=None, embed_size=self._E[i], dropout_rate=c['dropout_rate'],
                                 is_training=False, scope="prenet", in_features=cin)
                    out = CBHG(inputs=pre, embed_size=self._E[i], num_conv_banks=sd['num_conv_banks'],
                               num_highwaynet_blocks=sd['num_highwaynet_blocks'], dropout_rate=c['dropout_rate'],
                               is_training=False, scope="CBHG", use_Cudnn=c['use_Cudnn'], use_lstm=c['use_lstm'])
                    y = modules.dense(out, sd['n_output'], None, name="y_logits", out_f32=True)
                ys.append(y)
                cin = sd['n_output']
                if i + 1 < len(c['steps_v'][:2]):
                    x = y                                       # step2's input is y_mel (decoder.py:155); prenet converts on load
                    if blend:
                        f = float(self.f_mel_pred)
                        x = modules.axpby(y, f, target_mel, 1.0 - f)
        return ys[0], ys[1]

    def forward(self, x, target_mel=None):
        """x: encoder features [N, T, n_feat] (or posteriors [N, T, n_in] when there is no
        encoder).  Returns dict(y_mel, y_stft, y_phn) of float32 device tensors."""
        import torch
        n_in = self.cfg_d['input_shape'][-1]
        pad = modules._pad8(n_in)
        if self.encoder is not None:
            eo = self.encoder.forward(x, ppg_pad_to=pad, ppg_dtype=self.store.dtype)
            y_phn, ppg = eo['y_pred'], eo['ppg']
        else:
            y_phn = x
            ppg = torch.zeros((x.shape[0], x.shape[1], pad), dtype=torch.float32, device=x.device)
            ppg[:, :, :n_in] = x
            ppg = modules.convert(ppg, self.store.dtype)
        y_mel, y_stft = self.forward_from_ppg(ppg, target_mel)
        return {'y_mel': y_mel, 'y_stft': y_stft, 'y_phn': y_phn, 'dec_inputs': y_phn}

    # --------------------------------------------------------------------------- checkpoints
    def _all_variables(self):
        d = self.store.to_numpy()                    # the reference's Saver stores encoder vars too
        if self.encoder is not None:
            d.update(self.encoder.opt_state)
        d.update(self.opt_state)
        tr = getattr(self, '_trainer', None)
        if tr is not None:                            # Adam slots under TF's names (dec_opt/<var>/Adam[_1])
            d.update(tr.slot_dict())
        return d

    def save(self, save_path=None, i_checkpoint=None, verbose=True):
        """decoder.py:294-306."""
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
        """decoder.py:309-324."""
        if save_path is None:
            if i_checkpoint is None:
                save_path = tf_bundle.latest_checkpoint(self.cfg_d['model_path'])
            else:
                save_path = '{}/{}-{}'.format(self.cfg_d['model_path'], self.cfg_d['model_name'], int(i_checkpoint))
        try:
            w = tf_bundle.read_bundle(save_path, verify_crc=True)
            mine = [n for n in self.store.vars if n.startswith(self._scope + '/')]
            missing = [n for n in mine if n not in w]
            if missing:
                raise KeyError(missing[:3])
            self.store.load_dict({k: v for k, v in w.items() if k in self.store.vars}, strict=False)
            for k in self.opt_state:
                if k in w:
                    self.opt_state[k] = w[k]
            self.i_global_step = int(self.opt_state['dec_opt/global_step'])
            self.i_epoch = int(self.opt_state['dec_opt/epoch'])
            if self._F_NAME in self.store.vars:
                self.f_mel_pred = float(self.store.vars[self._F_NAME])
            if self.cfg_d['is_training']:
                # tf.train.Saver restores the Adam slots (dec_opt/<var>/Adam, Adam_1) and the step with the weights:
                # an existing trainer takes them now, a later one when it is created (_get_trainer)
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
    def _input_width(self):
        return self.encoder.cfg_d['input_shape'][-1] if self.encoder is not None else self.cfg_d['input_shape'][-1]

    def predict(self, x, batch_size=32, n_streams=2):
        """decoder.py:447-465: chunks of ``batch_size`` windows; returns the namedtuple
        (y_mel [N,T,n_mels], y_stft [N,T,n_stft], y_phn [N,T,n_in]) as numpy float32.
        The chunks are independent, so they are issued round-robin on ``n_streams`` HIP streams:
        one chunk's latency-bound recurrences overlap with another chunk's GEMMs (results are
        identical to the sequential order)."""
        import torch
        if self.cfg_d.get('use_target_mel_step2', False):
            raise Exception(' - ERROR, predict: a model built with use_target_mel_step2 needs target_mel for its second '
                            'stage (decoder.py:152); evaluate it through run() / exec_calc_metrics')
        n_chunks = (x.shape[0] + batch_size - 1) // batch_size
        use_streams = n_streams > 1 and n_chunks > 1 and torch.cuda.is_available()
        if use_streams and len(getattr(self, '_streams', None) or ()) != n_streams:
            self._streams = [torch.cuda.Stream() for _ in range(n_streams)]
        outs = []
        main = torch.cuda.current_stream() if torch.cuda.is_available() else None
        for k, i_s in enumerate(range(0, x.shape[0], batch_size)):
            x_batch = self._to_device(x[i_s:min(i_s + batch_size, x.shape[0])], self._input_width(), 'decoder input')
            if use_streams:
                st = self._streams[k % len(self._streams)]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    x_batch.record_stream(st)
                    outs.append(self.forward(x_batch))
            else:
                outs.append(self.forward(x_batch))
        if use_streams:
            for st in self._streams:
                main.wait_stream(st)
        y_mel_v, y_stft_v, y_phn_v = [], [], []
        for o in outs:
            y_mel_v.append(o['y_mel'].cpu().numpy())
            y_stft_v.append(o['y_stft'].cpu().numpy())
            y_phn_v.append(o['y_phn'].float().cpu().numpy())
        predict_nt = namedtuple('predict', 'y_mel y_stft y_phn')
        return predict_nt(np.concatenate(y_mel_v, axis=0), np.concatenate(y_stft_v, axis=0),
                          np.concatenate(y_phn_v, axis=0))

    def _losses(self, o, target_mel, target_stft):
        """decoder.py:185-199 on the host (scalars for eval; the training step fuses this)."""
        c = self.cfg_d
        tm = self._to_device(target_mel, c['steps_v'][0]['n_output'], 'target_mel')
        ts = self._to_device(target_stft, c['steps_v'][1]['n_output'], 'target_stft')
        mel_loss = float(c['mel_loss_weight'] * ((o['y_mel'] - tm) ** 2).mean())
        stft_loss = float(c['stft_loss_weight'] * ((o['y_stft'] - ts) ** 2).mean())
        if c['loss_type'] == 'log':
            loss = float(np.log(mel_loss) + np.log(stft_loss))
        elif c['loss_type'] == 'sum':
            loss = mel_loss + stft_loss
        else:
            raise Exception('- ERROR, _build_loss, loss_type not understood.')
        return np.float32(mel_loss), np.float32(stft_loss), np.float32(loss)

    def run(self, var, feed_dict={}):
        """decoder.py:468-469 for the handles this class defines."""
        single = not isinstance(var, (list, tuple))
        vs = [var] if single else list(var)
        if self.inputs not in feed_dict:
            raise Exception(' - ERROR, run: feed_dict must provide decoder.inputs')
        tm = None
        if self.cfg_d.get('use_target_mel_step2', False) and self.target_mel in feed_dict:
            tm = self._to_device(feed_dict[self.target_mel], self.cfg_d['steps_v'][0]['n_output'], 'target_mel')
        o = self.forward(self._to_device(feed_dict[self.inputs], self._input_width(), 'decoder input'), tm)
        losses = None
        res = []
        for v in vs:
            key = v.name.split('/')[-1]
            if key in ('mel_loss', 'stft_loss', 'loss'):
                if losses is None:
                    losses = dict(zip(('mel_loss', 'stft_loss', 'loss'),
                                      self._losses(o, feed_dict[self.target_mel], feed_dict[self.target_stft])))
                res.append(losses[key])
            elif key in o:
                res.append(o[key].float().cpu().numpy())
            else:
                raise Exception(' - ERROR, run: {} cannot be evaluated'.format(v))
        return res[0] if single else res

    def get_input_shape(self):
        """decoder.py:471-472."""
        return tuple(self.inputs.shape[1:])

    def exec_calc_metrics(self, inputs, target_mel, target_stft, summary_mode='validation'):
        """decoder.py:349-376 without the TensorBoard writers: (mel_loss, stft_loss, loss)."""
        if summary_mode not in ('train', 'validation', 'test'):
            raise Exception(' - ERROR, summary_mode={} not implemented'.format(summary_mode))
        c = self.cfg_d
        if c['is_training']:
            # the reference evaluates the graph it built (decoder.py:349-353): on a training model that is the
            # train-mode forward (dropout and batch statistics active; bn's updates_collections=None moves the
            # averages on every evaluation) -- the trainer's forward without the backward pass
            x = self._to_device(inputs, self._input_width(), 'decoder input')
            tm = self._to_device(target_mel, c['steps_v'][0]['n_output'], 'target_mel')
            ts = self._to_device(target_stft, c['steps_v'][1]['n_output'], 'target_stft')
            losses = self._get_trainer().forward_backward(x, tm, ts, backward=False)
            mel_loss, stft_loss = (np.float32(v) for v in losses.cpu().numpy())
            if c['loss_type'] == 'log':
                loss = np.float32(np.log(mel_loss) + np.log(stft_loss))
            elif c['loss_type'] == 'sum':
                loss = np.float32(mel_loss + stft_loss)
            else:
                raise Exception('- ERROR, _build_loss, loss_type not understood.')
            return mel_loss, stft_loss, loss
        tm = None
        if c.get('use_target_mel_step2', False):
            tm = self._to_device(target_mel, c['steps_v'][0]['n_output'], 'target_mel')
        o = self.forward(self._to_device(inputs, self._input_width(), 'decoder input'), tm)
        return self._losses(o, target_mel, target_stft)

    def eval_loss(self, ds_sampler, n_batchs=100):
        """decoder.py:474-493."""
        loss_v, mel_loss_v, stft_loss_v = [], [], []
        for i_batch, (mfcc_batch, mel_batch, stft_batch) in enumerate(ds_sampler):
            mel_loss, stft_loss, loss = self.exec_calc_metrics(mfcc_batch, mel_batch, stft_batch)
            loss_v.append(loss)
            mel_loss_v.append(mel_loss)
            stft_loss_v.append(stft_loss)
            print(' - i_batch={:2d} - loss={:0.3f}  -   mel_loss={:0.3f}  -   stft_loss={:0.3f}'.format(
                i_batch, np.mean(loss_v), np.mean(mel_loss_v), np.mean(stft_loss_v)))
        return np.mean(loss_v), np.mean(mel_loss_v), np.mean(stft_loss_v)

    # --------------------------------------------------------------------------- training
    def _get_trainer(self):
        if not self.cfg_d['is_training']:
            raise Exception('Model is not in training model')
        if getattr(self, '_trainer', None) is None:
            import training
            self._trainer = training.DecoderTrainer(self)
            if getattr(self, '_restored_ckpt', None) is not None:      # resume Adam state like tf.train.Saver
                self._trainer.resume(self._restored_ckpt)
                self._restored_ckpt = None
        return self._trainer

    def exec_train_step(self, inputs, target_mel, target_stft):
        """decoder.py:327-345: forward + backward + Adam on one batch (this rank's batch under data
        parallelism; gradients are averaged over ranks with an RCCL all-reduce).  Returns
        (mel_loss, stft_loss, loss, global_step, train_step) -- train_step is None (a TF op)."""
        import torch
        c = self.cfg_d
        tr = self._get_trainer()
        x = self._to_device(inputs, self._input_width(), 'decoder input')
        tm = self._to_device(target_mel, c['steps_v'][0]['n_output'], 'target_mel')
        ts = self._to_device(target_stft, c['steps_v'][1]['n_output'], 'target_stft')
        tr.forward_backward(x, tm, ts)
        world = torch.distributed.get_world_size() if (torch.distributed.is_available() and
                                                        torch.distributed.is_initialized()) else 1
        global_step = tr.apply_gradients(world)
        # (the losses were copied to the host behind the forward pass: no wait for the backward pass / Adam here; the next
        # step's launches queue behind them on the same stream)
        mel_loss, stft_loss = (np.float32(v) for v in tr.losses_on_host())
        if c['loss_type'] == 'log':
            loss = np.float32(np.log(mel_loss) + np.log(stft_loss))
        else:
            loss = np.float32(mel_loss + stft_loss)
        self.i_global_step = global_step
        return (mel_loss, stft_loss, loss, np.int32(global_step), None)

    def _set_f_mel_pred(self, value):
        self.f_mel_pred = float(value)
        if self._F_NAME in self.store.vars:
            self.store.vars[self._F_NAME].fill_(float(value))

    def _f_mel_pred_update(self):
        """decoder.py:258-260: f_mel_pred = min(1, 1.02 * tanh(epoch / target_mel_step2_val)) -- float32 like the
        graph's ops."""
        e = np.float32(int(self.opt_state['dec_opt/epoch']))
        f = np.minimum(np.float32(1.0), np.float32(1.02) * np.tanh(e / np.float32(self.cfg_d['target_mel_step2_val'])))
        self._set_f_mel_pred(np.float32(f))
        return self.f_mel_pred

    def _lr_decay(self):
        """decoder.py:248: lr = lr_start / (1 + decay * epoch)."""
        o = self.opt_state
        o['dec_opt/learning_rate'] = np.float32(float(o['dec_opt/learning_rate_start']) /
                                                (1.0 + float(o['dec_opt/learning_rate_decay']) * float(o['dec_opt/epoch'])))
        return o['dec_opt/learning_rate']

    def train(self):
        """decoder.py:379-444 (same prints and control flow; needs a dataset object providing
        get_n_windows / spec_window_sampler, which this package does not ship)."""
        add_pams = {}
        if 'ds_filter_d' in self.cfg_d.keys():
            add_pams['ds_filter_d'] = self.cfg_d['ds_filter_d']
        self.cfg_d['n_samples_trn'] = self.ds.get_n_windows(self.cfg_d['ds_prop_val'], **add_pams)[0]
        self.sampler_trn = self.ds.spec_window_sampler(batch_size=self.cfg_d['batch_size'], n_epochs=99999999,
                                                       randomize_samples=self.cfg_d['randomize_samples'],
                                                       sample_trn=True, prop_val=self.cfg_d['ds_prop_val'], **add_pams)
        self.sampler_val = self.ds.spec_window_sampler(batch_size=self.cfg_d['batch_size'], n_epochs=99999999,
                                                       randomize_samples=self.cfg_d['randomize_samples'],
                                                       sample_trn=False, prop_val=self.cfg_d['ds_prop_val'], **add_pams)
        self.cfg_d['n_steps_epoch_trn'] = self.cfg_d['n_samples_trn'] // self.cfg_d['batch_size']
        self.iter_val = iter(self.sampler_val)
        print(' Starting Training ...')
        print(' n_samples_trn:    ', self.cfg_d['n_samples_trn'])
        print(' n_steps_epoch_trn:', self.cfg_d['n_steps_epoch_trn'])
        print(' batch_size:       ', self.cfg_d['batch_size'])
        print(' n_epochs:         ', self.cfg_d['n_epochs'])
        input('Press --ENTER--')
        self.i_epoch = int(self.opt_state['dec_opt/epoch'])
        self.lr = self._lr_decay()
        for mfcc_trn, mel_trn, stft_trn in self.sampler_trn:
            mel_loss_trn, stft_loss_trn, loss_trn, global_step, train_step = self.exec_train_step(mfcc_trn, mel_trn, stft_trn)
            print(' - i_epoch={}   global_step={}   mel_loss_trn={:6.3f}  stft_loss_trn={:6.3f}  loss_trn={:6.3f}'.format(
                self.i_epoch, global_step, mel_loss_trn, stft_loss_trn, loss_trn))
            if (global_step / self.cfg_d['n_steps_epoch_trn']) % self.cfg_d['save_each_n_epochs'] == 0:
                print(' Saving, epoch={} ...'.format(self.i_epoch))
                self.save()
                mfcc_val, mel_val, stft_val = next(self.iter_val)
                mel_loss_val, stft_loss_val, loss_val = self.exec_calc_metrics(mfcc_val, mel_val, stft_val)
                print(' - i_epoch={}   global_step={}   mel_loss_val={:6.3f}   stft_loss_val={:6.3f}   loss_val={:6.3f}'.format(
                    self.i_epoch, int(global_step), mel_loss_val, stft_loss_val, loss_val))
            if global_step % self.cfg_d['n_steps_epoch_trn'] == 0:
                self.opt_state['dec_opt/epoch'] = np.int32(int(self.opt_state['dec_opt/epoch']) + 1)
                self.i_epoch = int(self.opt_state['dec_opt/epoch'])
                self.lr = self._lr_decay()
                if self.cfg_d.get('use_target_mel_step2', False):
                    self._f_mel_pred_update()
                    print(' - New epoch, f_mel_pred = {0:02f}'.format(self.f_mel_pred))
                if self.i_epoch >= self.cfg_d['n_epochs']:
                    break
        print(' End of Training !!!')
        return None
