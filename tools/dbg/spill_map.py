"""Where a build of K3 spills: scratch instructions per barrier-delimited region and per segment (priority markers).
usage: python tools/dbg/spill_map.py <extra -D flags...>"""
import subprocess, sys, re, collections
src = "/root/repo/hyperscanning_signal_analysis_amd/csrc/tf_inv.hip"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I/root/repo/hyperscanning_signal_analysis_amd/csrc",
       "-mllvm", "-simplifycfg-sink-common=false", "-mllvm", "-simplifycfg-hoist-common=false", "--cuda-device-only", "-S", src, "-o", "/tmp/isa/spill.s"] + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open("/tmp/isa/spill.s").read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN3hmv13tf_inv_kernelILi4ELb0EEEvNS_6TfArgsE:")][0]
region, seg = 0, 0
cnt = collections.Counter()
for i in range(start, len(lines)):
    t = lines[i].strip()
    if t.startswith(".Lfunc_end"): break
    if t.startswith("s_barrier"): region += 1; seg = 0
    if t.startswith("s_setprio"): seg += 1
    if t.startswith("scratch_"): cnt[(region, seg)] += 1
tot = sum(cnt.values())
print("total scratch instructions", tot)
for k in sorted(cnt): print(k, cnt[k])
