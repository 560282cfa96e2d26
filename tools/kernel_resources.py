"""Registers, LDS, scratch and static instruction counts of every kernel in csrc/mcpt_kernels.hip (cross-compiled for gfx950; no GPU
needed).  A guard against silent code-generation changes: +2 VGPRs on a kernel at 79 means 88 allocated and one resident wave less.
python tools/kernel_resources.py [extra -D flags]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "final-project-monte-carlo-path-tracer-with-microfacet-bsdf_amd", "csrc", "mcpt_kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"]

with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + ["-x", "hip", "-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
    t = open(out).read()
md = t[t.index("amdhsa.kernels"):]
print("%-46s %5s %5s %7s %8s %6s %7s %6s" % ("kernel", "vgpr", "sgpr", "lds B", "scratch", "spills", "instrs", "waves"), " (waves per SIMD allowed by the VGPRs alone)")
for blk in md.split("  - .agpr_count")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s*(\S+)", blk).group(1)
    name = g("name")
    i = t.index("\n" + name + ":")
    j = t.index("s_endpgm", i)
    n = sum(1 for l in t[i:j].split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";")))
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = short.replace("void ", "").replace("mcpt::(anonymous namespace)::", "").replace("mcpt::", "")
    short = re.sub(r"\(.*", "", short)
    v = int(g("vgpr_count"))
    alloc = (v + 7) // 8 * 8
    print("%-46s %5d %5s %7s %8s %6s %7d %6d" % (short, v, g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count"), n, min(8, 512 // max(alloc, 1))))
