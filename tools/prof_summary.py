"""Group a rocprofv3 *_kernel_stats.csv by kernel family: ms/step and launches/step."""
import collections
import csv
import re
import sys


def family(n: str) -> str:
    if any(t in n for t in ('gps_sa::', 'gps::', 'gps_attn::', 'gps_loss::', 'gps_ln::', 'gps_red::', 'gps_obj::')):
        return 'libgps_hip ' + re.sub(r'.*?(?:gps_sa::x3|gps_sa|gps_attn|gps_loss|gps_ln|gps_red|gps_obj|gps)::(\w+).*', r'\1', n)
    if 'BatchNorm' in n: return 'MIOpen BatchNorm'
    if 'max_pool' in n: return 'max_pool'
    if n.startswith('Cijk') and '_SB_' in n: return 'fp32 GEMM (rocBLAS/hipBLASLt)'
    if n.startswith('Cijk') or n.startswith('Custom_Cijk'): return 'bf16 GEMM (hipBLASLt)'
    if 'igemm' in n or 'miopenSp3' in n or 'Conv' in n: return 'MIOpen conv'
    if 'SoftMax' in n or 'softmax' in n: return 'softmax/logsoftmax'
    if 'attn_fwd' in n or 'bwd_kernel' in n: return 'SDPA flash (BERT)'
    if 'layer_norm' in n or 'GammaBeta' in n: return 'layernorm'
    if 'multi_tensor' in n: return 'optimizer (foreach)'
    if 'elementwise' in n or 'reduce_kernel' in n or 'Cat' in n or 'rocclr' in n or 'dropout' in n:
        return 'elementwise/copy/reduce'
    return 'other'


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    cat = collections.OrderedDict()
    tot = 0.0
    for r in rows:
        k = family(r['Name'])
        t = float(r['TotalDurationNs']) / 1e6 / steps
        cat.setdefault(k, [0.0, 0.0])
        cat[k][0] += t
        cat[k][1] += int(r['Calls']) / steps
        tot += t
    for k, (t, n) in sorted(cat.items(), key=lambda kv: -kv[1][0]):
        print(f"{t:8.3f} ms/step {n:8.1f} launches/step  {k}")
    print(f"{tot:8.3f} ms/step total kernel time ({steps} steps in the trace)")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7)
