"""Merges the two PMC passes of cfg2 (tools/pmc_summary.py outputs of the FETCH_SIZE and WRITE_SIZE runs) into profiles/rNN_cfg2_pmc_traffic.json,
the file bench.py reads `roofline.traffic` from.  usage: cfg2_pmc_traffic.py <fetch.json> <write.json> <out.json> <round>"""
import json
import sys
fe, wr, out, rnd = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3], sys.argv[4]
ALG = {"k_op_apply": 33998700, "k_cg_dirA": 29598864}
ker = {}
for k, v in fe.items():
    if k not in wr or "hbm_read_bytes_per_launch" not in v:
        continue
    r, w = v["hbm_read_bytes_per_launch"], wr[k].get("hbm_write_bytes_per_launch", 0.0)
    e = dict(launches=v["launches"], hbm_read_bytes_per_launch=int(r), hbm_write_bytes_per_launch=int(w), hbm_bytes_per_launch=float(int(r) + int(w)))
    if k in ALG:
        e["algorithmic_bytes_per_launch"] = ALG[k]
    ker[k] = e
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -f csv -- python bench.py --workload cfg2 --steps 2 --warmup 1 "
                     "--no-cpu-baseline --exact-launches (round %s tree); per-kernel sums / launches by tools/pmc_summary.py, merged by tools/cfg2_pmc_traffic.py" % rnd,
           "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; gfx950 correction per MI355X_MICROARCH.md (HBM section): read bytes = 2 * FETCH_SIZE * 1024, WRITE_SIZE as is",
           "kernels": ker}, open(out, "w"), indent=1)
print({k: v["hbm_bytes_per_launch"] for k, v in ker.items() if k in ("k_op_apply", "k_cg_dirA", "k_cg_upd")})
