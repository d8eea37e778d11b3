"""Instruction mix of the hottest loop of one kernel in a device assembly file (hipcc -S --offload-device-only).
usage: python tools/isa_mix.py /tmp/kr.s '<substring of the mangled kernel name>'
Prints the mnemonic histogram of the largest backward-branch loop and the whole kernel, and FLOP counts with FMA = 2, mul/add = 1."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(("; -- Begin function", ":")) or (l.startswith("_Z") and key in l.split(":")[0] and ":" in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
labels = {}
insts = []  # (index, mnemonic, text)
for l in body:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if not t or t.startswith((";", ".", "_Z")):
        continue
    insts.append((len(insts), t.split()[0], t))
loops = []
for i, mn, t in insts:
    if mn.startswith("s_cbranch") or mn == "s_branch":
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] <= i:
            loops.append((i - labels[tgt], labels[tgt], i))
loops.sort(reverse=True)


def hist(sub):
    h = collections.Counter(mn for _, mn, _ in sub)
    f64 = {k: v for k, v in h.items() if k.endswith("_f64") or "f64" in k}
    fma = sum(v for k, v in h.items() if k.split("_e")[0] in ("v_fma_f64", "v_fmac_f64"))
    muladd = sum(v for k, v in h.items() if k.split("_e")[0] in ("v_mul_f64", "v_add_f64"))
    valu = sum(v for k, v in h.items() if k.startswith("v_"))
    return h, f64, fma, muladd, valu


print(f"kernel: {body[0][:120]}")
print(f"instructions: {len(insts)}, loops (size, start, end): {loops[:4]}")
# A shared-rows kernel (kernels_modeb.hip) holds one body per wavefront of the workgroup ("part"), each with its own loop over the
# groups of the tile; every such loop contains the workgroup barrier.  One pass of ALL of them handles NP chunks of 64 observations,
# so the per-observation figures are the sums over the parts' loops divided by NP.
bar_loops = sorted({(a, b) for _, a, b in loops if any(mn == "s_barrier" for _, mn, _ in insts[a:b + 1])})
part_loops = [(a, b) for a, b in bar_loops if not any((a2 < a <= b2) or (a2 == a and b2 > b) for a2, b2 in bar_loops)]  # outermost
if len(part_loops) > 1:
    NP = len(part_loops)
    tot = collections.Counter()
    for a, b in part_loops:
        tot.update(mn for _, mn, _ in insts[a:b + 1])
    fma = sum(v for k, v in tot.items() if k.split("_e")[0] in ("v_fma_f64", "v_fmac_f64"))
    muladd = sum(v for k, v in tot.items() if k.split("_e")[0] in ("v_mul_f64", "v_add_f64"))
    valu = sum(v for k, v in tot.items() if k.startswith("v_"))
    lds = sum(v for k, v in tot.items() if k.startswith("ds_"))
    print(f"-- {NP} per-part group loops found (sizes {[b - a + 1 for a, b in part_loops]}; a part whose loop the assembler laid out without a "
          f"backward branch of its own is not found: the mean stands for it): per observation = mean over the parts: "
          f"{valu / NP:.1f} VALU, {lds / NP:.1f} LDS, fp64 FMA {fma / NP:.1f}, fp64 mul/add {muladd / NP:.1f}, "
          f"FLOP (FMA=2, mul/add=1) {(2 * fma + muladd) / NP:.1f}")
for name, sub in (("hottest loop", insts[loops[0][1]:loops[0][2] + 1] if loops else []), ("whole kernel", insts)):
    h, f64, fma, muladd, valu = hist(sub)
    print(f"-- {name}: {len(sub)} instructions, {valu} VALU, fp64 FMA {fma}, fp64 mul/add {muladd}, FLOP (FMA=2, mul/add=1) {2 * fma + muladd}")
    print("   " + ", ".join(f"{k} {v}" for k, v in h.most_common(28)))
