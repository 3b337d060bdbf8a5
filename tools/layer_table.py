"""Summarise a CANONSWAP_PROFILE_CSV file: per-family and per-layer time / TFLOP/s."""
import collections
import csv
import re
import sys


def grp(l):
    if re.match(r'(F|T)\.rb\d\.c\d|R\.s\d\.\d\.c\d', l): return '3D 32->32 convs'
    if re.match(r'T\.b\d\.c\d\.mask', l): return 'T mask convs'
    if re.match(r'T\.b\d\.c\d', l): return 'T fused blend convs'
    if re.match(r'G\..*\.n[01s]$', l): return 'G gamma/beta SPADE convs'
    if re.match(r'G\.shared', l): return 'G mlp_shared convs'
    if re.match(r'G\.', l): return 'G plain convs'
    if re.match(r'W\.(enc|dec)', l): return 'W hourglass enc/dec'
    if re.match(r'R\.rb2', l): return 'R 2D convs'
    return l


def main(path, detail):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r['ms']) for r in rows)
    agg = collections.OrderedDict()
    for r in rows:
        k = r['label'] if detail else grp(r['label'])
        a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r['ms']); a[2] += float(r['gflop'])
    print('total ms %.3f' % tot)
    for k, (n, ms, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: (70 if detail else 40)]:
        print('%-28s n=%3d  %7.3f ms  %5.1f%%  %8.1f GF  %7.1f TF/s' % (k, n, ms, 100 * ms / tot, gf, gf / ms if ms > 0 else 0))


if __name__ == '__main__':
    main(sys.argv[1], len(sys.argv) > 2)
