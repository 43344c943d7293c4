"""Timeline of ONE training step from a rocprofv3 kernel trace (--kernel-trace --output-format csv of
`bench.py --workload train`): kernel time by name, the launches longer than 0.25 ms in order with their stream, and
the idle stretches of the main stream.  python tools/train_timeline.py <..._kernel_trace.csv>"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Stream_Id']) for r in rows]
rows.sort(key=lambda r: r[1])
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
step = rows[adam[-2] + 1:adam[-1] + 1]


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)
    m = re.match(r'([A-Za-z0-9_:<>, ]+?)[(E]', n)
    return (m.group(1) if m else n)[:56]


t0 = step[0][1]
print('step: %.2f ms wall, %d launches, %.2f ms of kernel time' % ((step[-1][2] - t0) / 1e6, len(step), sum(r[2] - r[1] for r in step) / 1e6))
agg = collections.OrderedDict()
for r in step:
    d = agg.setdefault(short(r[0]), [0, 0.0])
    d[0] += 1
    d[1] += (r[2] - r[1]) / 1e6
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print('  %-58s %4d  %7.3f ms' % (n, v[0], v[1]))
streams = collections.Counter(r[3] for r in step)
main_id = streams.most_common(1)[0][0]
print('launches longer than 0.25 ms (start, duration, stream):')
for r in step:
    d = (r[2] - r[1]) / 1e6
    if d > 0.25:
        print('  %7.2f +%6.3f  %s%s  %s' % ((r[1] - t0) / 1e6, d, 's', r[3], short(r[0])))
main = [r for r in step if r[3] == main_id]
gaps = [((main[i + 1][1] - main[i][2]) / 1e6, i) for i in range(len(main) - 1)]
print('main stream %s: busy %.2f ms, idle %.2f ms' % (main_id, sum(r[2] - r[1] for r in main) / 1e6, sum(g for g, _ in gaps if g > 0)))
for g, i in sorted(gaps, reverse=True)[:8]:
    print('  idle %.3f ms at %.2f, after %s, before %s' % (g, (main[i][2] - t0) / 1e6, short(main[i][0]), short(main[i + 1][0])))
if len(sys.argv) > 2 and sys.argv[2] == 'gap':
    g, i = sorted(gaps, reverse=True)[0]
    lo, hi = main[i][2] - 400000, main[i + 1][1] + 300000
    print('around the longest idle stretch of the main stream:')
    for r in step:
        if r[2] >= lo and r[1] <= hi:
            print('  %8.3f +%6.3f  s%s  %s' % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3], short(r[0])))
