"""Opcode census of one kernel from the gfx950 assembly hipcc emits (-S --cuda-device-only): instruction classes per basic block of the
kernel body, the code-object notes (VGPRs, AGPRs, scratch, LDS, occupancy), s_waitcnt / s_nop counts.
usage: isa_census.py <file.s> <substring of the mangled kernel name> [min block size]"""
import collections, re, sys

path, want = sys.argv[1], sys.argv[2]
min_block = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l)
name = lines[start].rstrip(":")
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def cls(op):
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"): return "v_mad_64 (32x32+64)"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mul_u32", "v_mul_i32", "v_mad_u32", "v_mad_i32")): return "v_mul / v_mad 32"
    if op.startswith(("v_add", "v_sub", "v_addc", "v_subb", "v_subrev")): return "v_add / v_sub (incl. carry forms)"
    if op.startswith(("v_lshl", "v_lshr", "v_ashr", "v_alignbit", "v_bfe", "v_bfi", "v_and", "v_or", "v_xor", "v_not", "v_perm")): return "v shift / mask / logic"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane", "v_swap")): return "v_mov / accvgpr / lane"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"): return "v_cmp / v_cndmask"
    if op.startswith("v_"): return "v other"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")): return "s branch / call / barrier"
    if op.startswith("s_"): return "s other (scalar ALU, moves, exec masks)"
    if op.startswith(("global_load", "flat_load", "buffer_load")): return "global load"
    if op.startswith(("global_store", "flat_store", "buffer_store", "global_atomic")): return "global store / atomic"
    if op.startswith("scratch_"): return "scratch load / store"
    if op.startswith("ds_"): return "LDS"
    return "other"


blocks, cur, label = [], collections.Counter(), "entry"
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith((";", "//")): continue
    m = re.match(r"^(\.LBB\w+|\.L\w+):", s)
    if m:
        blocks.append((label, cur)); cur, label = collections.Counter(), m.group(1); continue
    if s.startswith("."): continue
    cur[cls(s.split()[0])] += 1
blocks.append((label, cur))
total = collections.Counter()
for _, c in blocks: total.update(c)
print(f"kernel {name}")
# code-object notes of this kernel (.amdhsa_ directives follow the body)
notes = {}
for l in lines[end:end + 400]:
    m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|private_segment_fixed_size|group_segment_fixed_size)\s+(\S+)", l)
    if m: notes[m.group(1)] = m.group(2)
    if l.startswith("_Z") or ".end_amdhsa_kernel" in l: break
for l in lines[end:]:
    m = re.match(r";\s*(NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs|codeLenInByte):\s*(\S+)", l.strip())
    if m and m.group(1) not in notes: notes[m.group(1)] = m.group(2)
    if l.startswith("_Z") and not l.startswith(name): break
print("notes: " + ", ".join(f"{k} = {v}" for k, v in notes.items()))
print(f"whole kernel: {sum(total.values())} instructions in {len(blocks)} basic blocks")
for k, v in sorted(total.items(), key=lambda kv: -kv[1]): print(f"    {v:6d}  {k}")
print(f"basic blocks of at least {min_block} instructions (the arithmetic of one mixed addition is straight-line code: its blocks carry the multiply-adds):")
for lab, c in blocks:
    n = sum(c.values())
    if n < min_block: continue
    vec = sum(v for k, v in c.items() if k.startswith("v"))
    print(f"  {lab:12s} {n:5d} instr, {vec:5d} vector, {c['v_mad_64 (32x32+64)']:5d} mad64 | " + ", ".join(f"{k.split(' (')[0]} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if not k.startswith("v_mad_64")))
