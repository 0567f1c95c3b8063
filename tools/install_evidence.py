"""Copy one evidence_round.sh session into profiles/ and print the figures the docs quote (one box, one session).
usage: python tools/install_evidence.py gpurun_out/evidence_r06 [--print-only]"""
import csv, json, os, shutil, sys

src = sys.argv[1]
R = "r06"
files = [f"{R}_bench_kernel_stats.csv", f"{R}_bench_pmc_summary.csv", f"{R}_bench_under_rocprof.json", f"{R}_bench_unprofiled.json",
         f"{R}_c5_byt5base_1m_e4m3.json", f"{R}_parity_margins.json", f"{R}_b1_latency_kernel_stats.txt", "pmc_traffic.json"]
if "--print-only" not in sys.argv:
    for f in files:
        shutil.copy(os.path.join(src, f), os.path.join("profiles", f))
d = json.load(open(os.path.join(src, f"{R}_bench_unprofiled.json")))
c = json.load(open(os.path.join(src, f"{R}_c5_byt5base_1m_e4m3.json")))
ks = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(src, f"{R}_bench_kernel_stats.csv")))}
def k(frag):
    return next(float(r["avg_us"]) for n, r in ks.items() if frag in n)
out = {
    "qps": d["value"], "ms": d["ms_per_step"], "probe": d["box_calibration"]["mfma_probe_tflops"], "copy": d["box_calibration"]["hbm_copy_gbs"],
    "wi_ms": d["roofline"]["avg_launch_ms"], "wi_tf": d["roofline"]["achieved"], "wi_frac": d["roofline"]["frac"], "wi_traffic": d["roofline"]["traffic"],
    "gemms": d["all_encoder_gemms_tflops"], "hbm": [(h["kernel"][:24], h["us"], h["frac"]) for h in d["roofline_hbm"]],
    "scan": {x: d["roofline_scan"][x] for x in ("frac", "mfma_frac", "whole_call_ms", "whole_call_frac", "whole_call_ms_back_to_back", "whole_call_frac_back_to_back", "traffic")},
    "shard": list(d["shard_call_us"].values()), "scan_qps": d["scan_only_qps"], "scan_qps8": d["scan_only_qps_e4m3_index"],
    "api": d["product_api_qps"], "reindex_s": d["reindex_130k"]["reindex_corpus_s"], "b1": d["b1_latency_ms"]["retrieve_wall_ms_by_state_bytes"],
    "b1_frac": d["roofline_b1"]["frac"], "train": {x: v.get("ms_per_step") for x, v in d["train_step"].items() if isinstance(v, dict)},
    "cpu": (d["cpu_baseline"]["value"], d["cpu_baseline"].get("premises_per_s")), "prem": d["premises_per_s"],
    "tiers": {x: v["premises_per_s"] for x, v in d["premises_per_s_by_length_tier"].items()},
    "rocprof_us": {"wi": k("EpiGegluBf16T"), "mixed": k("gemm_kernel_mixed"), "qkv": k("EpiStoreBf16T"), "att": k("attention_kernel"),
                   "filter": k("sim_filter_kernel"), "sample": k("sim_scan_kernel"), "gsel": k("gather_select_kernel"), "sel": k("select_kernel<"),
                   "embed": k("embed_copy_kernel")},
    "c5": {"qps": c["value"], "ms": c["ms_per_step"], "wi_frac": c["roofline"]["frac"], "scan_ms": c["roofline_scan"]["ms_per_step"],
           "scan_frac": c["roofline_scan"]["frac"], "call_ms": c["roofline_scan"]["whole_call_ms"], "mfma": c["roofline_scan"]["mfma_tflops"]},
    "hash": json.load(open(os.path.join(src, "pmc_traffic.json")))["kernel_source_hash"],
}
print(json.dumps(out, indent=1))
