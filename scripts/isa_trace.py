"""Compact instruction trace of the MFMA-densest basic block of a kernel: M=mfma r=ds_read w=ds_write
G=global_load v=valu s=salu [..]=s_waitcnt.  usage: isa_trace.py file.hip mangled_name_substring"""
import re, subprocess, sys
src, key = sys.argv[1], sys.argv[2]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", src, "-o", "/tmp/_t.s", "--cuda-device-only"], stderr=subprocess.DEVNULL, check=True)
s = open("/tmp/_t.s").read()
for m in re.finditer(r"^(_Z\S*%s\S*):" % re.escape(key), s, re.M):
    i = m.start(); body = s[i:s.index(".Lfunc_end", i)]
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
    best = max(blocks, key=lambda b: b.count("v_mfma"))
    seq = []
    for l in best.split("\n"):
        l = l.strip()
        if not l or l[0] in ";.": continue
        op = l.split()[0]
        if op.startswith("v_mfma"): seq.append("M")
        elif op.startswith("ds_read"): seq.append("r")
        elif op.startswith("ds_write"): seq.append("w")
        elif op.startswith("global_load") or op.startswith("buffer_load"): seq.append("G")
        elif op.startswith("s_waitcnt"): seq.append("[" + l.split(None, 1)[1].replace(" ", "") + "]")
        elif op.startswith("s_barrier"): seq.append("|BAR|")
        elif op.startswith("v_"): seq.append("v")
        elif op.startswith("s_"): seq.append("s")
        else: seq.append("?")
    print(m.group(1)[:70]); print("".join(seq)); print()
