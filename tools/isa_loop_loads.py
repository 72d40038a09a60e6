"""Which kernels issue byte / halfword global loads INSIDE a loop?  (tools/isa_loop_loads.py x.s, x.s from `hipcc --cuda-device-only -S`.)

A sub-dword load of a wave-uniform address is a VECTOR load on gfx9 (there is no s_load_ubyte / s_load_ushort); the value is needed as
a scalar, so the compiler follows it with s_waitcnt vmcnt(0) -- which also waits for every tile prefetch issued before it.  Round 4
found two GEMM loops running at one memory latency per K-tile because of this (net_gemm.h, wg_ctx)."""
import re, sys, subprocess, shutil
lines = open(sys.argv[1]).read().split("\n")
funcs, cur = {}, None
for l in lines:
    m = re.match(r"^(_Z[A-Za-z0-9_]+):", l)
    if m:
        cur = m.group(1); funcs[cur] = []
    elif cur is not None:
        funcs[cur].append(l)
filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
def dem(n):
    if not filt: return n
    return subprocess.run([filt, n], capture_output=True, text=True).stdout.strip()
for name, body in funcs.items():
    inloop, hits = False, []
    for l in body:
        if l.startswith(".LBB"):
            inloop = "Loop" in l
        if inloop and re.search(r"global_load_(ubyte|sbyte|ushort|sshort)", l):
            hits.append(l.strip().split()[0])
    if hits:
        print(len(hits), sorted(set(hits)), dem(name)[:170])
