#!/usr/bin/env python3
"""Static instruction mix of one kernel of a built translation unit (no GPU):  python tools/isa_stats.py echo echo_range_sl_kernel ILi1ELi1ELi4E
Counts by class (VALU fp64 / fp32 / transcendental / integer multiply / other VALU, MFMA, LDS, vector memory, scalar, waits) + register budget."""
import collections, os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from test_isa_cpu import CodeObject

unit, parts = sys.argv[1], sys.argv[2:]
co = CodeObject(tempfile.mkdtemp(), unit)
name, meta, asm = co.find(*parts)
cls = collections.Counter()
for ln in asm:
    op = ln.split()[0]
    if op.startswith("v_mfma"): c = "mfma"
    elif re.match(r"v_(log|exp|sin|cos|sqrt|rsq|rcp)_", op): c = "valu transcendental"
    elif re.match(r"v_(mad_u64_u32|mul_hi_u32|mul_lo_u32|mad_i64_i32)", op): c = "valu int multiply (quarter rate)"
    elif "_f64" in op: c = "valu fp64"
    elif "_f32" in op: c = "valu fp32"
    elif op.startswith("v_"): c = "valu other"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c = "vmem"
    elif op.startswith("s_waitcnt"): c = "s_waitcnt"
    elif op.startswith("s_barrier"): c = "s_barrier"
    elif op.startswith("s_"): c = "salu"
    else: c = "other"
    cls[c] += 1
print(name)
print({k: meta[k] for k in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size") if k in meta})
for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
    print(f"  {k:36s} {v}")
print(f"  {'total':36s} {len(asm)}")
if os.environ.get("ISA_OPS"):
    ops = collections.Counter(ln.split()[0] for ln in asm)
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:int(os.environ["ISA_OPS"])]:
        print(f"    {k:32s} {v}")
if os.environ.get("ISA_DUMP"):
    open(os.environ["ISA_DUMP"], "w").write("\n".join(asm) + "\n")
