"""tools/isa_classes.py <kernel.s> — static VALU instruction mix of one kernel's assembly (tools/isa_dump.sh writes it) by issue-cost class
(tools/microbench/vgpr_banks.hip, SIMD-cycles per wave64 instruction at 7 waves per SIMD): plain 2.3-2.5; compare / select / carry /
min-max / 3-input integer / 64-bit / SGPR-operand 4.1-4.3; lane ops (readlane, writelane, readfirstlane, permute, dpp moves); transcendental 8.4."""
import collections
import re
import sys

slow_prefix = ("v_cmp", "v_cndmask", "v_addc", "v_subb", "v_min", "v_max", "v_med3", "v_add3", "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3",
               "v_xad", "v_bfe", "v_bfi", "v_alignbit", "v_mad_u64", "v_mad_i64", "v_mul_hi", "v_mul_lo", "v_mbcnt", "v_bcnt", "v_add_co", "v_sub_co",
               "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_perm")
lane_prefix = ("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_mov_b32_dpp", "ds_bpermute", "ds_permute", "ds_swizzle")
trans_prefix = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
counts = collections.Counter()
by_op = collections.Counter()
for line in open(sys.argv[1]):
    line = line.split(";")[0].strip()
    m = re.match(r"(v_\w+|ds_bpermute\w*|ds_permute\w*|ds_swizzle\w*)\b(.*)", line)
    if not m:
        continue
    op, rest = m.group(1), m.group(2)
    if op.startswith(lane_prefix):
        cls = "lane"
    elif op.startswith(trans_prefix):
        cls = "transcendental"
    elif op.startswith(slow_prefix):
        cls = "compare / select / carry / min-max / 3-input / 64-bit"
    elif re.search(r"(?<![\w\[])s\d+|s\[\d+:\d+\]|\bvcc\b|\bexec\b", rest) and not op.startswith("v_cndmask"):
        cls = "SGPR operand"
    else:
        cls = "plain"
    counts[cls] += 1
    by_op[(cls, op.split("_e32")[0].split("_e64")[0])] += 1
total = sum(counts.values())
weights = {"plain": 2.4, "SGPR operand": 4.2, "compare / select / carry / min-max / 3-input / 64-bit": 4.3, "lane": 4.3, "transcendental": 8.4}
print("%d VALU-side instructions" % total)
for cls, n in counts.most_common():
    print("  %5d  %4.1f %%  %s" % (n, 100.0 * n / total, cls))
print("  weighted mean issue cost %.2f SIMD-cycles per instruction" % (sum(weights[c] * n for c, n in counts.items()) / total))
for cls in counts:
    if cls != "plain":
        top = [(op, n) for (c, op), n in by_op.most_common() if c == cls][:8]
        print("  %s: %s" % (cls, ", ".join("%s %d" % t for t in top)))
