"""Extracts the reference's own saved graph (enc_14_ckpt/encoder-136512.meta, a serialized TF-1.9
MetaGraphDef: a DATA file of the reference) into tests/golden/enc_14_graph.json: for every node of
the forward graph and of the optimizer's update ops its name, op, inputs and plain attributes
(strings, ints, floats, bools, dtypes, int lists, small constant values).  tests/test_graph_pins_cpu.py
checks the oracle's TF semantics against it (padding rules, batch-norm epsilon / decay, pooling window,
GRU cell wiring, dropout keep probability, Adam constants, argmax type).

Pure protobuf WIRE decoding (varints and length-delimited fields), nothing from the file is executed.
Run in the build container only (/root/reference is not on the GPU box):
    python tools/make_graph_fixture.py [/root/reference]"""
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def varint(b, i):
    v = s = 0
    while True:
        c = b[i]; i += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, i
        s += 7


def fields(b):
    """Yield (field number, wire type, value) of one message; value = int or bytes."""
    i, n = 0, len(b)
    while i < n:
        key, i = varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 2:
            ln, i = varint(b, i)
            v = b[i:i + ln]; i += ln
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError('wire type %d' % wt)
        yield f, wt, v


def sint64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


DTYPES = {1: 'float32', 2: 'float64', 3: 'int32', 7: 'string', 9: 'int64', 10: 'bool', 20: 'resource'}


def packed_ints(v, wt):
    if wt == 0:
        return [sint64(v)]
    out, i = [], 0
    while i < len(v):
        x, i = varint(v, i)
        out.append(sint64(x))
    return out


def tensor_value(b):
    """TensorProto -> {'dtype', 'shape', 'value' (only when <= 8 elements)}."""
    dtype, shape, content, fl, il = None, [], None, [], []
    for f, wt, v in fields(b):
        if f == 1:
            dtype = DTYPES.get(v, v)
        elif f == 2:
            for f2, _, v2 in fields(v):
                if f2 == 2:
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:
                            shape.append(sint64(v3))
        elif f == 4:
            content = v
        elif f == 5:
            fl += [struct.unpack('<f', v)[0]] if wt == 5 else list(struct.unpack('<%df' % (len(v) // 4), v))
        elif f == 7:
            il += packed_ints(v, wt)
    n = 1
    for d in shape:
        n *= d
    out = {'dtype': dtype, 'shape': shape}
    if n <= 8:
        if content is not None and dtype in ('float32', 'int32', 'int64'):
            fmt = {'float32': 'f', 'int32': 'i', 'int64': 'q'}[dtype]
            out['value'] = list(struct.unpack('<%d%s' % (len(content) // struct.calcsize(fmt), fmt), content))
        elif fl:
            out['value'] = fl
        elif il:
            out['value'] = il
    return out


def attr_value(b):
    for f, wt, v in fields(b):
        if f == 2:
            return v.decode('utf-8', 'replace')
        if f == 3:
            return sint64(v)
        if f == 4:
            return struct.unpack('<f', v)[0]
        if f == 5:
            return bool(v)
        if f == 6:
            return DTYPES.get(v, v)
        if f == 8:
            return tensor_value(v)
        if f == 7:                                   # TensorShapeProto
            dims = []
            for f2, _, v2 in fields(v):
                if f2 == 2:
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:
                            dims.append(sint64(v3))
            return dims
        if f == 1:                                   # ListValue
            out = []
            for f2, wt2, v2 in fields(v):
                if f2 == 2:
                    out.append(v2.decode('utf-8', 'replace'))
                elif f2 == 3:
                    out += packed_ints(v2, wt2)
                elif f2 == 4:
                    out += [struct.unpack('<f', v2)[0]] if wt2 == 5 else list(struct.unpack('<%df' % (len(v2) // 4), v2))
                elif f2 == 6:
                    out += [DTYPES.get(x, x) for x in packed_ints(v2, wt2)]
            return out
    return None


def node(b):
    d = {'name': None, 'op': None, 'input': [], 'attr': {}}
    for f, wt, v in fields(b):
        if f == 1:
            d['name'] = v.decode()
        elif f == 2:
            d['op'] = v.decode()
        elif f == 3:
            d['input'].append(v.decode())
        elif f == 5:
            k = val = None
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    k = v2.decode()
                elif f2 == 2:
                    val = attr_value(v2)
            if k not in ('_class', '_output_shapes') and val is not None:
                d['attr'][k] = val
    return d


def extract(ref='/root/reference'):
    meta = open(os.path.join(ref, 'enc_14_ckpt', 'encoder-136512.meta'), 'rb').read()
    info, nodes, producer = {}, [], None
    for f, wt, v in fields(meta):
        if f == 1:                                   # MetaInfoDef
            for f2, _, v2 in fields(v):
                if f2 == 5:
                    info['tensorflow_version'] = v2.decode()
                elif f2 == 6:
                    info['tensorflow_git_version'] = v2.decode()
        elif f == 2:                                 # GraphDef
            for f2, _, v2 in fields(v):
                if f2 == 1:
                    nodes.append(node(v2))
                elif f2 == 4:
                    for f3, _, v3 in fields(v2):
                        if f3 == 1:
                            producer = v3
    keep = [n for n in nodes if 'gradients' not in n['name'] and 'summar' not in n['name'].lower()
            and not n['name'].startswith('save') and 'Initializer' not in n['name']
            and n['op'] not in ('Assign', 'Fill', 'NoOp', 'ScalarSummary', 'MergeSummary')]
    return {'source': 'enc_14_ckpt/encoder-136512.meta', 'meta_info': info, 'graph_producer': producer,
            'n_nodes_total': len(nodes), 'nodes': keep}


def main():
    out = extract(sys.argv[1] if len(sys.argv) > 1 else '/root/reference')
    dst = os.path.join(ROOT, 'tests', 'golden', 'enc_14_graph.json')
    json.dump(out, open(dst, 'w'), separators=(',', ':'), sort_keys=True)
    print('%d nodes total, %d kept -> %s (%d bytes)' % (out['n_nodes_total'], len(out['nodes']), dst, os.path.getsize(dst)))


if __name__ == '__main__':
    main()
