"""Static instruction mix of one kernel from hipcc -S output.
usage: python tools/isa_mix.py <file.hip> <mangled-name-substring> [...]"""
import collections, os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(HERE, "..", "pde_opt_amd", "csrc", sys.argv[1])
out = "/tmp/_isa.s"
subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I/opt/rocm/include", "-fno-slp-vectorize", "-S",
                "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for pat in sys.argv[2:]:
    for si, s in enumerate(starts):
        name = lines[s].split(":")[0]
        if pat not in name:
            continue
        end = next((i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
        ops = collections.Counter()
        for l in lines[s:end]:
            m = re.match(r"\s+([a-z_0-9]+)\s", l)
            if m:
                ops[m.group(1)] += 1
        g = collections.Counter()
        for k, v in ops.items():
            if k.startswith("v_pk"): g["v_pk_*"] += v
            elif re.match(r"v_(log|exp|rcp|sqrt|rsq)", k): g[k] += v
            elif k.startswith("v_div"): g["v_div_*"] += v
            elif k.startswith("ds_"): g[k] += v
            elif k.startswith(("global_", "buffer_")): g[k] += v
            elif k.startswith(("s_barrier", "s_waitcnt")): g[k] += v
            elif k.startswith(("v_cmp", "v_cndmask")): g["cmp/cndmask"] += v
            elif k.startswith(("v_fma", "v_fmac")): g["fma"] += v
            elif re.match(r"v_(mul|add|sub)_f", k): g["mul/add/sub_f"] += v
            elif k.startswith("v_mov"): g["v_mov"] += v
            elif k.startswith("v_"): g["other VALU"] += v
            elif k.startswith("s_"): g["SALU"] += v
        tot = sum(v for k, v in ops.items() if k.startswith("v_"))
        print(name[:70], "| static VALU", tot)
        print("   ", ", ".join(f"{k}={v}" for k, v in sorted(g.items(), key=lambda x: -x[1])))
