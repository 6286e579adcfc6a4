"""Per-basic-block instruction counts of one kernel in a -save-temps .s file (VALU, SALU, VMEM, LDS) with landmarks."""
import re, sys
path, start, end = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lines = open(path).read().split("\n")[start - 1:end]
blocks, cur = [], {"name": "entry", "line": start, "v": 0, "s": 0, "m": 0, "d": 0, "marks": set(), "br": []}
for i, l in enumerate(lines):
    t = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", t)
    if m:
        blocks.append(cur)
        cur = {"name": m.group(1), "line": start + i, "v": 0, "s": 0, "m": 0, "d": 0, "marks": set(), "br": []}
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur["v"] += 1
        for k in ("v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_cvt_f32_ubyte", "v_pk_fma_f32", "v_div_fmas_f64", "v_rcp_f32", "v_readlane", "v_writelane", "v_mbcnt"):
            if op.startswith(k):
                cur["marks"].add(k[2:])
    elif op.startswith("s_"):
        cur["s"] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["br"].append(t.split()[-1])
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        cur["m"] += 1
        if op.startswith("scratch_"): cur["marks"].add("scratch")
        if "atomic" in op: cur["marks"].add("atomic")
    elif op.startswith("ds_"):
        cur["d"] += 1
blocks.append(cur)
tv = sum(b["v"] for b in blocks)
print("total valu", tv, "salu", sum(b["s"] for b in blocks), "vmem", sum(b["m"] for b in blocks), "lds", sum(b["d"] for b in blocks))
for b in blocks:
    print(f'{b["name"]:12s} L{b["line"]:5d} v{b["v"]:4d} s{b["s"]:4d} m{b["m"]:3d} d{b["d"]:3d} {",".join(sorted(b["marks"]))} -> {" ".join(b["br"])}')
