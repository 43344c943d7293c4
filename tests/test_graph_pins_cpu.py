"""Pins the oracle's TensorFlow-side semantics to the reference's OWN saved graph.

The reference ships no tests and TensorFlow cannot be imported here, so no activations of the
reference exist to compare against ("parity unpinned").  What the reference does
ship is the MetaGraphDef its Saver wrote next to the trained weights
(/root/reference/enc_14_ckpt/encoder-136512.meta, TF 1.9.0-rc0): tests/golden/enc_14_graph.json is that
graph's forward and optimizer-update nodes (names, ops, inputs, attributes, small constants; data only,
written by tools/make_graph_fixture.py with a protobuf wire decoder).  These tests check, without a GPU:

  * every attribute the restatement depends on (convolution padding / strides / layout, no conv bias,
    batch-norm epsilon and moving-average decay, pooling window, dropout keep probability, Adam
    constants, argmax type, variable names and shapes);
  * the GRU cell and the highway block NODE BY NODE: a tiny interpreter evaluates the saved sub-graphs
    on random tensors and the oracle's functions must give the same numbers -- the gate order
    (reset first), `u*h + (1-u)*c`, and `H*T + x*(1-T)` are thereby read from the reference's graph,
    not from a reading of TensorFlow's documentation.
What stays outside the graph (TensorFlow kernels' own arithmetic: how SAME distributes an odd padding,
FusedBatchNorm's variance formula) is marked [ext] in SURVEY.md and covered by the oracle's own tests."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle import model_oracle as mo


@pytest.fixture(scope='module')
def graph():
    g = json.load(open(os.path.join(GOLDEN, 'enc_14_graph.json')))
    g['by_name'] = {n['name']: n for n in g['nodes']}
    return g


def _ops(graph, op, prefix='encoder/'):
    return [n for n in graph['nodes'] if n['op'] == op and n['name'].startswith(prefix)]


def _const(graph, name):
    v = graph['by_name'][name]['attr']['value']['value']
    return v[0] if len(v) == 1 else v


class _Interp:
    """Evaluates a sub-graph of the fixture on torch tensors.  ``feeds``: node name -> tensor for the
    sub-graph's boundary (placeholders, loop variables, variables, pre-computed Tensordot results)."""
    UNARY = {'Relu': torch.relu, 'Sigmoid': torch.sigmoid, 'Tanh': torch.tanh, 'Identity': lambda x: x, 'Enter': lambda x: x}
    BINARY = {'Mul': torch.mul, 'Sub': torch.sub, 'Add': torch.add, 'BiasAdd': torch.add, 'MatMul': torch.matmul,
              'RealDiv': torch.div, 'SquaredDifference': lambda a, b: (a - b) ** 2, 'Equal': torch.eq}

    def __init__(self, graph, feeds):
        self.N, self.memo, self.visited = graph['by_name'], dict(feeds), []

    def get(self, ref):
        name, _, idx = ref.partition(':')
        v = self.node(name)
        return v[int(idx or 0)] if isinstance(v, (list, tuple)) else v

    def node(self, name):
        if name in self.memo:
            return self.memo[name]
        n = self.N[name]
        self.visited.append(n['op'])
        ins = [i for i in n['input'] if not i.startswith('^')]
        op = n['op']
        if op == 'Const':
            v = n['attr']['value']
            out = torch.tensor(v['value'], dtype=torch.float64 if v['dtype'] == 'float32' else torch.int64).reshape(v['shape'])
        elif op in self.UNARY:
            out = self.UNARY[op](self.get(ins[0]))
        elif op in self.BINARY:
            assert not n['attr'].get('transpose_a') and not n['attr'].get('transpose_b')
            out = self.BINARY[op](self.get(ins[0]), self.get(ins[1]))
        elif op == 'ConcatV2':
            out = torch.cat([self.get(i) for i in ins[:-1]], dim=int(self.get(ins[-1])))
        elif op == 'Split':                              # inputs: (split_dim, value)
            out = list(torch.chunk(self.get(ins[1]), n['attr']['num_split'], dim=int(self.get(ins[0]))))
        else:
            raise AssertionError('unexpected op %s at %s' % (op, name))
        self.memo[name] = out
        return out


def test_saved_by_the_pinned_tensorflow(graph):
    assert graph['meta_info']['tensorflow_version'] == '1.9.0-rc0'
    assert graph['n_nodes_total'] == 2701 and graph['graph_producer'] == 26


def test_variables_are_the_checkpoints_and_the_oracles(graph):
    import tf_bundle
    cfg = json.load(open(os.path.join(ROOT, 'speech-cloner_amd', 'hp', 'encoder_cfg_d.json')))
    in_graph = {n['name']: tuple(n['attr']['shape']) for n in _ops(graph, 'VariableV2')}
    in_oracle = {name: tuple(shape) for name, shape, _ in mo.model_variable_shapes(cfg, 'encoder')}
    assert in_graph == in_oracle and len(in_graph) == 38
    ckpt = {k: tuple(e.shape) for k, e in tf_bundle.list_bundle(os.path.join(GOLDEN, 'enc_14_ckpt', 'encoder-136512')).items()}
    for k, shp in in_graph.items():
        assert ckpt[k] == shp, k
    # one Adam update per trainable tensor: everything except the 6 moving statistics
    upd = [n for n in graph['nodes'] if n['op'] == 'ApplyAdam']
    assert sorted(n['input'][0] for n in upd) == sorted(k for k in in_graph if 'moving_' not in k)
    assert sorted(mo.trainable_names(in_oracle)) == sorted(n['input'][0] for n in upd)


def test_convolutions_norms_and_pooling(graph):
    convs = _ops(graph, 'Conv2D')
    assert len(convs) == 8                                               # 6 bank widths + 2 projections
    for n in convs:
        a = n['attr']
        assert (a['padding'], a['data_format'], a['strides'], a['dilations']) == ('SAME', 'NHWC', [1, 1, 1, 1], [1, 1, 1, 1])
        assert n['input'][1].endswith('ExpandDims_1')                    # filter = the variable, H axis of size 1 added
    # tf.layers.conv1d(use_bias=False): no bias variable under any convolution scope
    assert not [v for v in _ops(graph, 'VariableV2') if '/conv1d/' in v['name'] and v['name'].endswith('bias')]
    norms = _ops(graph, 'FusedBatchNorm')
    assert len(norms) == 3
    for n in norms:
        assert n['attr']['epsilon'] == mo.BN_EPS == float(np.float32(1e-3)) and n['attr']['data_format'] == 'NHWC'
    decays = [n for n in graph['nodes'] if n['op'] == 'Const' and n['name'].endswith('/decay') and 'AssignMovingAvg' in n['name']]
    assert len(decays) == 6
    for n in decays:                                                     # moving -= (moving - batch) * (1 - 0.999)
        assert n['attr']['value']['value'][0] == float(np.float32(1.0 - mo.BN_DECAY))
    (pool,) = _ops(graph, 'MaxPool')
    assert (pool['attr']['ksize'], pool['attr']['strides'], pool['attr']['padding']) == ([1, 1, 2, 1], [1, 1, 1, 1], 'SAME')
    # the oracle's reading of those attributes: window 2 / stride 1 / SAME pads one frame at the END
    x = torch.randn(2, 7, 5, dtype=torch.float64)
    ref = torch.maximum(x, torch.cat([x[:, 1:], x[:, -1:]], 1))
    assert torch.equal(mo.max_pool_2_same(x), ref)
    # ... and SAME for an even filter width k: (k-1)//2 zeros in front, the rest behind
    k, cin, cout = 4, 3, 2
    w = torch.randn(k, cin, cout, dtype=torch.float64)
    xp = torch.cat([x.new_zeros(2, 1, 5), x, x.new_zeros(2, 2, 5)], 1)[:, :, :cin]
    ref = torch.stack([sum(xp[:, t + j] @ w[j] for j in range(k)) for t in range(7)], 1)
    assert torch.allclose(mo.conv1d(x[:, :, :cin], w), ref, atol=1e-12)
    import modules
    assert (float(np.float32(modules.BN_EPS)), modules.BN_DECAY) == (mo.BN_EPS, mo.BN_DECAY)   # the product's constants too (passed as float32)


def test_projections_are_wired_as_the_oracle_reads_them(graph):
    N = graph['by_name']
    # conv1d_1 -> norm -> relu ; conv1d_2 -> norm (no activation) ; + prenet output (after dropout)
    assert N['encoder/CBHG/Relu']['input'] == ['encoder/CBHG/Squeeze'] and N['encoder/CBHG/Squeeze']['input'] == ['encoder/CBHG/conv1d_1_1/Identity']
    assert N['encoder/CBHG/add']['input'] == ['encoder/CBHG/Squeeze_1', 'encoder/prenet/dropout2/dropout/mul']
    assert N['encoder/CBHG/Squeeze_1']['input'] == ['encoder/CBHG/conv1d_2_1/Identity']
    # the bank's norm sees the concatenation of the 6 widths; relu follows; the pool feeds conv1d_1
    bn_in = N[N['encoder/CBHG/conv1d_banks/bn/FusedBatchNorm']['input'][0]]
    assert bn_in['op'] == 'ExpandDims'
    # modules.py:157-162: outputs = concat((outputs, conv_k), -1) for k = 2..K  ->  channel blocks in the order k = 1..6
    order, cur = [], N[bn_in['input'][0]]
    while cur['op'] == 'ConcatV2':
        assert len(cur['input']) == 3 and _const(graph, cur['input'][2]) == -1
        order.append(cur['input'][1])
        cur = N[cur['input'][0]]
    order.append(cur['name'])
    want = ['encoder/CBHG/conv1d_banks/%sconv1d/conv1d/conv1d/Squeeze' % ('' if k == 1 else 'num_%d/' % k) for k in range(6, 0, -1)]
    assert order == want
    (pool,) = _ops(graph, 'MaxPool')
    relu = N[N[pool['input'][0]]['input'][0]]
    assert relu['op'] == 'Relu' and N[relu['input'][0]]['op'] == 'Squeeze'    # relu(bn(banks)) -> pool
    # outputs (encoder.py:109-111): softmax of the logits, argmax cast to int32
    assert N['encoder/y_pred_class']['op'] == 'Cast' and N['encoder/y_pred_class']['attr']['DstT'] == 'int32'
    assert N['encoder/y_pred_class']['input'] == ['encoder/ArgMax'] and N['encoder/ArgMax']['input'][0] == 'encoder/y_logits/BiasAdd'


@pytest.mark.parametrize('direction', ['fw', 'bw'])
def test_gru_cell_node_by_node(graph, direction):
    H, C, B = 40, 40, 5
    rng = np.random.RandomState(3)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s))
    scope = 'encoder/CBHG/gru/bidirectional_rnn/%s' % direction
    loop = '%s/%s/while/' % (scope, direction)
    w = {scope + '/gru_cell/gates/kernel': t(C + H, 2 * H), scope + '/gru_cell/gates/bias': t(2 * H),
         scope + '/gru_cell/candidate/kernel': t(C + H, H), scope + '/gru_cell/candidate/bias': t(H)}
    x = t(B, 2, C)
    # oracle: two steps from the zero state, forward in time
    ref = mo.gru_direction(x, w, scope, reverse=False)
    h = torch.zeros(B, H, dtype=torch.float64)
    for step in range(2):
        feeds = {k + '/read': v for k, v in w.items()}
        feeds[loop + 'TensorArrayReadV3'] = x[:, step]                   # the loop's input at this step
        feeds[loop + 'Identity_3'] = h                                   # the loop-carried state
        it = _Interp(graph, feeds)
        h = it.get(loop + 'gru_cell/add')
        assert torch.allclose(h, ref[:, step], atol=1e-12), step
    assert sorted(set(it.visited)) == ['Add', 'BiasAdd', 'ConcatV2', 'Const', 'Enter', 'MatMul', 'Mul', 'Sigmoid', 'Split', 'Sub', 'Tanh']
    # the state the loop carries is what it emits: the cell's output IS the next state
    N = graph['by_name']
    assert any(n['op'] == 'NextIteration' and n['input'] == [loop + 'gru_cell/add'] for n in graph['nodes'])
    assert N[loop + 'gru_cell/sub/x']['attr']['value']['value'] == [1.0]


def test_highway_block_node_by_node(graph):
    H, B = 40, 6
    rng = np.random.RandomState(4)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s))
    s = 'encoder/CBHG/highwaynet_0'
    w = {s + '/dense1/kernel': t(H, H), s + '/dense1/bias': t(H), s + '/dense2/kernel': t(H, H), s + '/dense2/bias': t(H)}
    x = t(B, 3, H)
    feeds = {'encoder/CBHG/add': x, s + '/dense1/bias/read': w[s + '/dense1/bias'], s + '/dense2/bias/read': w[s + '/dense2/bias'],
             # tf.layers.dense on a rank-3 tensor = Tensordot over the last axis (reshape + MatMul + reshape)
             s + '/dense1/Tensordot': x @ w[s + '/dense1/kernel'], s + '/dense2/Tensordot': x @ w[s + '/dense2/kernel']}
    N = graph['by_name']
    for d in ('dense1', 'dense2'):
        mm = N['%s/%s/Tensordot/MatMul' % (s, d)]
        assert not mm['attr']['transpose_a'] and not mm['attr']['transpose_b']
    out = _Interp(graph, feeds).get(s + '/add')
    assert torch.allclose(out, mo.highwaynet(x, w, s), atol=1e-12)
    assert N[s + '/sub/x']['attr']['value']['value'] == [1.0]


def test_dropout_and_adam_constants(graph):
    cfg = json.load(open(os.path.join(ROOT, 'speech-cloner_amd', 'hp', 'encoder_cfg_d.json')))
    for i in (1, 2):
        kp = _const(graph, 'encoder/prenet/dropout%d/dropout/keep_prob' % i)
        assert kp == float(np.float32(1.0 - cfg['dropout_rate']))
    # tf.layers.dropout: x / keep_prob * floor(keep_prob + U[0,1))
    N = graph['by_name']
    assert N['encoder/prenet/dropout1/dropout/mul']['input'] == ['encoder/prenet/dropout1/dropout/div', 'encoder/prenet/dropout1/dropout/Floor']
    assert N['encoder/prenet/dropout1/dropout/div']['op'] == 'RealDiv'
    b1, b2, eps = (_const(graph, 'opt/Adam/' + k) for k in ('beta1', 'beta2', 'epsilon'))
    assert (b1, b2, eps) == tuple(float(np.float32(v)) for v in (0.9, 0.999, 1e-8))
    import inspect
    d = inspect.signature(mo.adam_step).parameters
    assert (d['beta1'].default, d['beta2'].default, d['eps'].default) == (0.9, 0.999, 1e-8)
    assert (cfg['beta1'], cfg['beta2'], cfg['epsilon']) == (0.9, 0.999, 1e-8)
    for n in graph['nodes']:
        if n['op'] == 'ApplyAdam':
            assert n['attr']['use_nesterov'] is False
            assert n['input'][3:9] == ['opt/beta1_power/read', 'opt/beta2_power/read', 'opt/learning_rate/read',
                                       'opt/Adam/beta1', 'opt/Adam/beta2', 'opt/Adam/epsilon']


def test_loss_metrics_and_learning_rate_schedule(graph):
    N = graph['by_name']
    rng = np.random.RandomState(5)
    logits = torch.from_numpy(rng.standard_normal((2, 9, 61)))
    target = torch.eye(61, dtype=torch.float64)[torch.from_numpy(rng.randint(0, 61, (2, 9)))]
    # loss (encoder.py:134-137): Mean over every frame of softmax_cross_entropy_with_logits(logits, target)
    xe = N['loss/softmax_cross_entropy_with_logits']
    assert xe['op'] == 'SoftmaxCrossEntropyWithLogits'
    assert N[xe['input'][0]]['input'][0] == 'encoder/y_logits/BiasAdd' and N[xe['input'][1]]['input'][0] == 'encoder/target'
    mean = N['loss/cross_entropy']
    assert mean['op'] == 'Mean' and mean['input'][0] == 'loss/softmax_cross_entropy_with_logits/Reshape_2' and _const(graph, mean['input'][1]) == [0, 1]   # over windows and frames
    per_frame = -(target * torch.log_softmax(logits, -1)).sum(-1)       # the op's definition
    assert torch.allclose(mo.encoder_loss(logits, target), per_frame.mean(), atol=1e-12)
    # metrics (encoder.py:143-150): evaluated from the saved nodes
    pred = torch.softmax(logits, -1)
    it = _Interp(graph, {'encoder/y_pred': pred, 'encoder/target': target,
                         'metric/predictions': pred.argmax(-1), 'metric/labels': target.argmax(-1)})
    acc_g = it.get('metric/accuracy/Equal').double().mean()
    mse_g = it.get('metric/SquaredDifference').mean()
    assert N['metric/accuracy/Mean']['input'][0] == 'metric/accuracy/Cast' and N['metric/mean_squared_error']['input'][0] == 'metric/SquaredDifference'
    assert N['metric/predictions']['input'] == ['metric/ArgMax_1'] and N['metric/ArgMax_1']['input'][0] == 'encoder/y_pred'
    assert N['metric/labels']['input'] == ['metric/ArgMax'] and N['metric/ArgMax']['input'][0] == 'encoder/target'
    acc, mse = mo.encoder_metrics(logits, target)
    assert torch.allclose(acc, acc_g) and torch.allclose(mse, mse_g, atol=1e-15)
    # learning-rate schedule (encoder.py:183): lr_start / (1 + decay * epoch), from the saved nodes
    assert N['opt/Cast']['input'] == ['opt/epoch/read']
    for epoch in (0, 3, 14):
        it = _Interp(graph, {'opt/learning_rate_start/read': torch.tensor(1e-3, dtype=torch.float64),
                             'opt/learning_rate_decay/read': torch.tensor(1e-3, dtype=torch.float64),
                             'opt/Cast': torch.tensor(float(epoch), dtype=torch.float64)})
        assert abs(float(it.get('opt/truediv')) - 1e-3 / (1.0 + 1e-3 * epoch)) < 1e-18
    # ... which the trained checkpoint's own scalars satisfy (epoch 14)
    import tf_bundle
    sc = tf_bundle.read_bundle(os.path.join(GOLDEN, 'enc_14_ckpt', 'encoder-136512'))
    lr = float(sc['opt/learning_rate_start']) / (1.0 + float(sc['opt/learning_rate_decay']) * float(sc['opt/epoch']))
    assert abs(float(sc['opt/learning_rate']) - lr) < 1e-9


def test_fixture_is_the_reference_graph(graph, reference_dir):
    """Provenance (build container only; skipped where /root/reference is absent): the committed fixture is
    exactly what tools/make_graph_fixture.py extracts from the reference's .meta file."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_graph_fixture', os.path.join(ROOT, 'tools', 'make_graph_fixture.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = json.loads(json.dumps(mod.extract(reference_dir)))
    assert fresh['nodes'] == graph['nodes'] and fresh['meta_info'] == graph['meta_info']


# ------------------------------------------------------------------------------------------------------
# The WHOLE forward graph, as the reference saved it (training mode: dropout, batch statistics), evaluated
# on the reference's trained weights and compared with the oracle's forward.  What the interpreter below
# supplies itself are TensorFlow's primitive kernels ([ext] in SURVEY.md: SAME convolution / pooling
# arithmetic, FusedBatchNorm on batch statistics, Tensordot = matmul over the last axis, softmax); every
# connection between them -- which tensor feeds which layer, dropout placement, concat orders, the residual,
# time reversal of the backward recurrence, the loop-carried state -- is read from the saved graph.
class _FullInterp(_Interp):
    def __init__(self, graph, feeds, weights):
        super().__init__(graph, feeds)
        self.w = weights

    def const(self, name):
        v = self.N[name]['attr']['value']
        return v['value'] if 'value' in v else None

    def node(self, name):
        if name in self.memo:
            return self.memo[name]
        n = self.N[name]
        op, ins = n['op'], [i for i in n['input'] if not i.startswith('^')]
        out = None
        if op == 'VariableV2':
            out = self.w[name]
        elif op == 'Reshape' and name.endswith('/Tensordot'):
            # tf.layers.dense on [B, T, C]: tensordot(x, kernel, [[2], [0]]) lowered to transpose/reshape/MatMul/reshape
            scope = name
            assert self.const(scope + '/axes') == [2]
            x = self.get(self.N[scope + '/transpose']['input'][0])
            k = self.get(self.N[scope + '/transpose_1']['input'][0])
            mm = self.N[scope + '/MatMul']
            assert not mm['attr']['transpose_a'] and not mm['attr']['transpose_b']
            out = x @ k
        elif op == 'ExpandDims':
            out = self.get(ins[0]).unsqueeze(int(self.const(ins[1])[0]))
        elif op == 'Squeeze':
            out = self.get(ins[0]).squeeze(n['attr']['squeeze_dims'][0])
        elif op == 'Conv2D':                             # [B, 1, T, C] * [1, k, C, O], NHWC, stride 1, SAME
            a = n['attr']
            assert (a['padding'], a['data_format'], a['strides']) == ('SAME', 'NHWC', [1, 1, 1, 1])
            x, k = self.get(ins[0]), self.get(ins[1])
            assert x.shape[1] == 1 and k.shape[0] == 1
            out = mo.conv1d(x[:, 0], k[0]).unsqueeze(1)
        elif op == 'FusedBatchNorm':                     # training mode: batch statistics over (B, H, W)
            assert n['attr']['is_training'] is True
            x, g_, b_ = self.get(ins[0]), self.get(ins[1]), self.get(ins[2])
            mean = x.mean(dim=(0, 1, 2))
            var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
            cnt = x.numel() // x.shape[-1]                # outputs 1, 2: batch mean, UNBIASED batch variance ([ext])
            out = [(x - mean) * torch.rsqrt(var + n['attr']['epsilon']) * g_ + b_, mean, var * (cnt / (cnt - 1.0))]
        elif op == 'MaxPool':
            assert (n['attr']['ksize'], n['attr']['strides'], n['attr']['padding']) == ([1, 1, 2, 1], [1, 1, 1, 1], 'SAME')
            x = self.get(ins[0])
            out = torch.maximum(x, torch.cat([x[:, :, 1:], x[:, :, -1:]], 2))
        elif op == 'ReverseV2':
            out = torch.flip(self.get(ins[0]), dims=[int(v) for v in self.const(ins[1])])
        elif op == 'Transpose':                          # dynamic_rnn: batch-major <-> time-major
            out = self.get(ins[0]).transpose(0, 1)
        elif op == 'TensorArrayGatherV3':                # the while loop: iterate the saved cell sub-graph
            base = name[:name.index('/TensorArrayStack')]            # .../bidirectional_rnn/fw/fw
            loop = base + '/while/'
            seq = self.get(base + '/transpose')                      # [T, B, C], what the loop's TensorArray was filled with
            scat = self.N[base + '/TensorArrayUnstack/TensorArrayScatter/TensorArrayScatterV3']
            assert scat['input'][2] == base + '/transpose'
            cell = base.rsplit('/', 1)[0] + '/gru_cell'               # variables live under .../fw/gru_cell
            h = torch.zeros(seq.shape[1], self.w[cell + '/candidate/bias'].shape[0], dtype=seq.dtype)
            outs = []
            for t in range(seq.shape[0]):
                feeds = {loop + 'TensorArrayReadV3': seq[t], loop + 'Identity_3': h}
                for v in ('gates/kernel', 'gates/bias', 'candidate/kernel', 'candidate/bias'):
                    feeds[cell + '/' + v + '/read'] = self.w[cell + '/' + v]
                h = _Interp({'by_name': self.N}, feeds).get(loop + 'gru_cell/add')
                outs.append(h)
            out = torch.stack(outs, 0)
        elif op == 'Softmax':
            out = torch.softmax(self.get(ins[0]), -1)
        elif op == 'Reshape' and name == 'encoder/Reshape':
            x = self.get(ins[0]); out = x.reshape(-1, x.shape[-1])
        elif op == 'Reshape' and name == 'encoder/y_pred':
            out = self.get(ins[0]).reshape(self.get('encoder/y_logits/BiasAdd').shape)
        if out is None:
            return super().node(name)
        self.memo[name] = out
        return out


def test_whole_forward_graph_on_the_trained_weights(graph):
    import tf_bundle
    cfg = json.load(open(os.path.join(ROOT, 'speech-cloner_amd', 'hp', 'encoder_cfg_d.json')))
    w = mo.to_torch({k: v for k, v in tf_bundle.read_bundle(os.path.join(GOLDEN, 'enc_14_ckpt', 'encoder-136512')).items()
                     if k.startswith('encoder/')}, torch.float64)
    g = np.load(os.path.join(GOLDEN, 'encoder_fwd.npz'))
    x = torch.from_numpy(g['x'][:2, :96]).double()                  # 2 windows x 96 frames of real front-end features
    rng = np.random.RandomState(6)
    m1 = torch.from_numpy((rng.rand(2, 96, 80) < 0.6).astype(np.float64))
    m2 = torch.from_numpy((rng.rand(2, 96, 40) < 0.6).astype(np.float64))
    it = _FullInterp(graph, {'encoder/inputs': x, 'encoder/prenet/dropout1/dropout/Floor': m1,
                             'encoder/prenet/dropout2/dropout/Floor': m2}, w)
    logits_g = it.get('encoder/y_logits/BiasAdd')
    pred_g = it.get('encoder/y_pred')
    taps = {}
    stats = {}
    logits, pred, cls, out = mo.encoder_forward(x, w, cfg, taps=taps, is_training=True, masks=(m1, m2), stats_out=stats)
    # moving statistics after the step (updates_collections=None: AssignSub of (moving - batch) * decay)
    for var_scope, op_scope in (('encoder/CBHG/conv1d_banks/bn', 'encoder/CBHG/conv1d_banks/bn'),
                                ('encoder/CBHG/conv1d_1', 'encoder/CBHG/conv1d_1_1'), ('encoder/CBHG/conv1d_2', 'encoder/CBHG/conv1d_2_1')):
        for stat, upd in (('moving_mean', 'AssignMovingAvg'), ('moving_variance', 'AssignMovingAvg_1')):
            asg = graph['by_name']['%s/%s' % (op_scope, upd)]
            assert asg['op'] == 'AssignSub' and asg['input'][0] == '%s/%s' % (var_scope, stat)
            new = w['%s/%s' % (var_scope, stat)] - it.get(asg['input'][1])
            assert torch.allclose(new, stats['%s/%s' % (var_scope, stat)], atol=1e-12), (var_scope, stat)
    assert torch.allclose(it.get('encoder/prenet/dropout2/dropout/mul'), taps['prenet'], atol=1e-12)
    assert torch.allclose(it.get('encoder/CBHG/highwaynet_0/add'), taps['highway'], atol=1e-10)
    assert torch.allclose(it.get('encoder/CBHG/gru/concat'), out, atol=1e-10)
    assert torch.allclose(logits_g, logits, atol=1e-9) and torch.allclose(pred_g, pred, atol=1e-10)
    assert torch.equal(logits_g.argmax(-1).to(torch.int32), cls)
    # the interpreter really walked the graph: every layer type was visited
    assert {'Conv2D', 'FusedBatchNorm', 'MaxPool', 'ReverseV2', 'TensorArrayGatherV3', 'Softmax'} <= {
        graph['by_name'][k]['op'] for k in it.memo if k in graph['by_name']}
