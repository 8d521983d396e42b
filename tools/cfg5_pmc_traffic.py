"""Merges the three PMC passes of cfg5 (tools/pmc_summary.py outputs of the FETCH_SIZE, WRITE_SIZE and MFMA runs) into profiles/rNN_cfg5_pmc_traffic.json,
the file bench.py reads the `traffic` / `mfma_busy_frac_pmc` fields of the cfg5 rooflines from.
usage: cfg5_pmc_traffic.py <fetch.json> <write.json> <mfma.json> <out.json> <round>"""
import json
import sys
fe, wr, mf, out, rnd = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), json.load(open(sys.argv[3])), sys.argv[4], sys.argv[5]
ker = {}
for k, v in fe.items():
    if k not in wr or "hbm_read_bytes_per_launch" not in v:
        continue
    r, w = v["hbm_read_bytes_per_launch"], wr[k].get("hbm_write_bytes_per_launch", 0.0)
    e = dict(launches=v["launches"], hbm_read_bytes_per_launch=int(r), hbm_write_bytes_per_launch=int(w), hbm_bytes_per_launch=float(int(r) + int(w)))
    if k in mf and "mfma_busy_frac" in mf[k] and mf[k].get("SQ_INSTS_VALU_MFMA_F64", 0) > 0:
        e["mfma_busy_frac"] = round(mf[k]["mfma_busy_frac"], 4)
        e["mfma_f64_flop_per_launch"] = mf[k]["mfma_f64_flop_per_launch"]
    ker[k] = e
dirm = [k for k in ker if k.startswith("k_cg_dirM<") and k.endswith("false>")]
pair = None
if dirm and "k_cg_upd<false>" in ker:
    pair = dict(kernels=[dirm[0], "k_cg_upd<false>"], hbm_bytes_per_krylov_iteration=ker[dirm[0]]["hbm_bytes_per_launch"] + ker["k_cg_upd<false>"]["hbm_bytes_per_launch"],
                algorithmic_bytes_per_krylov_iteration=13491324.0,
                note="every launch re-fetches its operands through the fabric: the eight L2s are invalidated at kernel boundaries, the 8.5 MB operator and the {r, u} "
                     "records gathered by all eight XCDs come from the Infinity Cache / HBM side in every Krylov iteration")
json.dump({"source": "rocprofv3 --pmc <group> --kernel-trace -f csv -- python bench.py --workload cfg5 --steps 3 --warmup 2 --no-cpu-baseline --no-float32 --no-prewarm "
                     "--no-variants; one run per counter group (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64), summed per kernel by "
                     "tools/pmc_summary.py, merged by tools/cfg5_pmc_traffic.py (round %s tree; the <.., true> instantiations are the opt-in Jacobi-PCG leg of the same run)" % rnd,
           "units": "FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction per MI355X_MICROARCH.md (HBM section): read bytes = 2 * FETCH_SIZE * 1024, WRITE_SIZE as is; "
                    "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)",
           "krylov_iteration_pair": pair, "kernels": ker}, open(out, "w"), indent=1)
print(pair, {k: (v["hbm_bytes_per_launch"], v.get("mfma_busy_frac")) for k, v in ker.items() if "symm_gemm" in k})
