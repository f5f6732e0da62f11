#!/usr/bin/env python
"""profiles/traffic.json["emd_forward"] from the per-kernel PMC summaries of one mvp_emd_forward call
(tools/profile_round.sh + tools/pmc_summary.py):  python tools/make_traffic.py profiles/r3_pmc_ <out.json>"""
import glob, hashlib, json, os, subprocess, sys
prefix, outp = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()   # (bench.py: emd_sources_digest -- the same recipe)
for f in sorted(glob.glob(os.path.join(ROOT, "mvp_benchmark_amd", "csrc", "emd*"))):
    h.update(os.path.basename(f).encode())
    h.update(open(f, "rb").read())
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    head = None
kernels = ["emd_auction_kernel", "emd_lean_kernel", "emd_lean_tiers_kernel"]
labels = ["emd_auction_kernel<4> (rounds 0..~100)", "emd_lean_kernel<4> (to round 300)",
          "emd_lean_tiers_kernel (the rest, 8 .. 2 workgroups per cloud by load)"]
per, tot = {}, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_VALU": 0.0, "SQ_INSTS_SALU": 0.0, "SQ_WAIT_ANY": 0.0, "SQ_WAVE_CYCLES": 0.0}
for k in kernels:
    d = json.load(open(prefix + k + ".json"))
    c = d["counters"]
    dur = sorted(d["duration_ms"])[len(d["duration_ms"]) // 2] if d["duration_ms"] else None
    per[k] = {"duration_ms": dur, "FETCH_SIZE_KB": c.get("FETCH_SIZE"), "WRITE_SIZE_KB": c.get("WRITE_SIZE"),
              "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU")}
    for t in tot:
        tot[t] += c.get(t, 0.0)
try:
    doc = json.load(open(outp))
except Exception:
    doc = {}
doc["emd_forward"] = {
    "batch": 64, "points": 16384, "eps": 0.004, "iters": 3000, "kernels": labels,
    "FETCH_SIZE_KB": tot["FETCH_SIZE"], "WRITE_SIZE_KB": tot["WRITE_SIZE"],
    "SQ_INSTS_VALU": tot["SQ_INSTS_VALU"], "SQ_INSTS_SALU": tot["SQ_INSTS_SALU"],
    "wait_any_frac": tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"] if tot["SQ_WAVE_CYCLES"] else None,
    "per_kernel": per,
    "emd_sources_sha256": h.hexdigest(), "taken_at": {"git_head": head, "prefix": prefix},
    "source": "%s{emd_auction_kernel,emd_lean_kernel,emd_lean_tiers_kernel}.json (rocprofv3 --pmc, "
              "separate passes with --kernel-trace only, per launch, summed over the three kernels of one mvp_emd_forward call; "
              "same-XCD stores; tools/profile_round.sh + tools/pmc_summary.py + tools/make_traffic.py)" % prefix,
    "note": "L2<->fabric bytes incl. Infinity-Cache hits; FETCH_SIZE is NOT doubled (the gfx950 1/2-count applies to 16-B/lane streaming "
            "reads; these kernels issue scattered 4-16 B accesses) -- uncalibrated.  The instruction counts include the polling loops "
            "of the cluster barriers (most of the SALU).",
}
json.dump(doc, open(outp, "w"), indent=1)
print(json.dumps(doc["emd_forward"], indent=1)[:1500])
