#!/usr/bin/env python3
"""Disassembly helpers for the kernels of csrc/engine.hip (no GPU needed): compile with --save-temps into /tmp/cimg_kasm and
print, per kernel, instruction counts by class, spill traffic (v_readlane / v_writelane / scratch) and the s_waitcnt vmcnt(0)
count; `--dump <kernel>` writes that kernel's assembly to /tmp/cimg_kasm/<kernel>.s.   usage: tools/kasm.py [--dump name] [hipcc flags]"""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/cimg_kasm"
args = sys.argv[1:]
dump = None
if args[:1] == ["--dump"]:
    dump, args = args[1], args[2:]
os.makedirs(out, exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-mllvm",
                       "-structurizecfg-skip-uniform-regions=true", "--save-temps", *args, "-c",
                       os.path.join(ROOT, "compressed-image_amd", "csrc", "engine.hip"), "-o", "engine.o"], cwd=out, stderr=subprocess.DEVNULL)
text = open(os.path.join(out, "engine-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
kern, cur = collections.OrderedDict(), None
for line in text:
    m = re.match(r"^(cimg_\w+):", line)
    if m:
        cur = m.group(1); kern[cur] = []; continue
    if cur and re.match(r"^\.Lfunc_end\d+:", line):
        cur = None; continue
    if cur is not None:
        kern[cur].append(line)
print("%-30s %7s %6s %6s %6s %6s %6s %8s %8s %7s" % ("kernel", "instr", "valu", "salu", "ds", "vmem", "branch", "readlane", "writelane", "vmcnt0"))
for k, lines in kern.items():
    ins = [l.strip() for l in lines if re.match(r"^\s+[a-z]", l) and not l.strip().startswith((".", ";"))]
    c = collections.Counter()
    for i in ins:
        op = i.split()[0]
        if op.startswith("v_readlane"): c["rl"] += 1
        elif op.startswith("v_writelane"): c["wl"] += 1
        if op.startswith("v_"): c["valu"] += 1
        elif op.startswith(("s_cbranch", "s_branch")): c["br"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        if op == "s_waitcnt" and "vmcnt(0)" in i: c["vm0"] += 1
    print("%-30s %7d %6d %6d %6d %6d %6d %8d %8d %7d" % (k, len(ins), c["valu"], c["salu"], c["ds"], c["vmem"], c["br"], c["rl"], c["wl"], c["vm0"]))
    if dump == k:
        open(os.path.join(out, k + ".s"), "w").write("\n".join(lines))
        print("  ->", os.path.join(out, k + ".s"))
