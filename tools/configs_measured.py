"""profiles/<tag>_configs_measured.jsonl from the round's evidence: tools/run_config.py lines (gpurun_out/configs_<tag>.jsonl) plus the
PMC summaries under profiles/ (tools/summarize_prof.py).  Per configuration and build: kernel ms, Mrays/s, the HBM roofline
(ALGORITHMIC bytes = 4 B x pixels written + 4 B x texel and skybox fetches + the scene's wire bytes; traffic = FETCH_SIZE + WRITE_SIZE,
KiB counters) and the VALU-issue roofline ((VALU x 2 + transcendental x 8 cycles) / 1024 SIMDs / 2.4 GHz).
   python tools/configs_measured.py r02"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
SCENE_BYTES = {"c2": 723, "ref800": 723, "c5strip": 723, "c3": 4 + 64 * 96 + 4 + 96 + 4 + 3 * 48, "c4": 4 + 10000 * 96 + 4 + 96 + 4 + 3 * 48}

def pmc(cfg, build):
    path = os.path.join(ROOT, "profiles", f"{tag}_{cfg}_rocprof_summary.md")
    if not os.path.exists(path):
        return {}
    out, on = {}, False
    for line in open(path):
        if line.startswith("### kernel"):
            on = f"wt_{build}::" in line
        elif on and re.match(r"^[A-Za-z_0-9]+,[0-9.]+,\d+$", line.strip()):
            k, v, _ = line.strip().split(",")
            out[k] = float(v)
    return out

rows = []
for line in open(os.path.join(ROOT, "gpurun_out", f"configs_{tag}.jsonl")):
    d = json.loads(line)
    c, build = d["counters"], "strict" if d["strict"] else "fast"
    cfg = d["config"]
    alg = 4 * d["pixels"] + 4 * (c["texel_fetches"] + c["sky_fetches"]) + SCENE_BYTES[cfg]
    row = dict(config=cfg, frame=d["frame"], depth=d["depth"], build=build, kernel_ms=d["kernel_ms"], Mrays_s=d["Mrays_s"],
               rays_per_px=d["rays_per_px"], lane_util=d["lane_util"],
               rays_traced_fraction=round((c["segments"] + c["shadow_rays_traced"]) / max(c["segments"] + c["shadow_rays"], 1), 4),
               roofline_hbm=dict(algorithmic_bytes=alg, achieved_GBps=round(alg / d["kernel_ms"] / 1e6, 1), peak_GBps=8000.0,
                                 frac=round(alg / d["kernel_ms"] / 1e6 / 8000.0, 5)))
    m = pmc(cfg, build)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        row["roofline_hbm"]["traffic_bytes"] = int((m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    if "SQ_INSTS_VALU" in m and "SQ_INSTS_VALU_TRANS_F32" in m:
        v, t = m["SQ_INSTS_VALU"], m["SQ_INSTS_VALU_TRANS_F32"]
        floor_ms = ((v - t) * 2.0 + t * 8.0) / 1024 / 2.4e9 * 1e3
        row["roofline_valu_issue"] = dict(valu_instructions=int(v), transcendental=int(t), issue_floor_ms=round(floor_ms, 4),
                                          frac=round(floor_ms / d["kernel_ms"], 4),
                                          note=f"(VALU x 2 + transcendental x 8 cycles) / 1024 SIMDs / 2.4 GHz, counters from profiles/{tag}_{cfg}_rocprof_summary.md")
    rows.append(row)
with open(os.path.join(ROOT, "profiles", f"{tag}_configs_measured.jsonl"), "w") as f:
    for r in rows:
        f.write(json.dumps(r) + "\n")
        print(json.dumps(r)[:230])
