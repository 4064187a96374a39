"""Per-kernel-family HBM traffic and SQ counters from the rocprofv3 --pmc passes of scripts/pmc_collect.sh ->
profiles/r03_pmc_traffic_b<B>_d<D>.json (bench.py reads `families[*].hbm_bytes_per_launch` into roofline.traffic).

    python scripts/pmc_traffic_summary.py gpurun_out/pmc_r03 128 32 profiles/r03_pmc_traffic_b128_d32.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads, so it is doubled
(MI355X_MICROARCH.md, section HBM).  Averages are over ALL dispatches of a kernel, including the launches of the update
kernels that find no work and exit at once -- the same convention as bench.py's algorithmic_bytes_per_launch."""
import collections
import csv
import glob
import json
import os
import sys

src, B, D, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
FAMILY = [("k_flush", "flush"), ("k_qr_apply_reg", "qr_apply"), ("k_qr_apply", "qr_apply"), ("k_bmult_chain", "bmult"), ("k_update_gather", "gather"),
          ("k_zgemm<2, 2", "gemm"), ("k_zgemm<1, 1", "gemm_small"), ("k_update_decide", "decide"), ("k_qr_panel", "qr_panel"), ("k_trsm_block", "trsm"),
          ("k_udt_init", "udt_init"), ("k_lu_", "lu")]


def family(name):
    import re
    m = re.search(r"k_zgemm<(\d+), (\d+), (?:true|false), (\d+)", name)        # <TM, TN, M3, TAG, OPA, OPB>
    if m and m.group(3) == "1":
        return "gemm_in_factorisation"          # TAG = 1: LU trailing updates, triangular solves, block Gram-Schmidt products (kernels_gemm.hip)
    if re.search(r"k_flush<(?:true|false), (?:true|false), \d+, 1>", name) or "k_flush_lds<1>" in name:
        return "lu_update"                      # <M3, FULL, STAGE, TAG = 1>: trailing updates of the LU factorisation on the flush kernel (kernels_lu.hip)
    for key, fam in FAMILY:
        if key in name:
            return fam
    return "other"


def collect(pass_name):
    files = glob.glob(os.path.join(src, pass_name, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in files:
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[fam][r["Counter_Name"]] += 1
    return acc, cnt


fa, fc = collect("fetch")
wa, wc = collect("write")
sa, sc = collect("sq")
la, lc = collect("lds")
fams = {}
for fam in sorted(set(fa) | set(wa)):
    nf, nw = fc[fam].get("FETCH_SIZE", 0), wc[fam].get("WRITE_SIZE", 0)
    if not nf or not nw:
        continue
    rd = 2.0 * 1024.0 * fa[fam]["FETCH_SIZE"] / nf
    wr = 1024.0 * wa[fam]["WRITE_SIZE"] / nw
    e = {"dispatches": nf, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
         "note": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, mean over all %d dispatches, %d chains per launch, delaySteps %d"
                 % (nf, B, D)}
    s = sa.get(fam)
    if s and s.get("SQ_WAVE_CYCLES"):
        wcyc = s["SQ_WAVE_CYCLES"]
        e["sq"] = {"wait_any_frac": s.get("SQ_WAIT_ANY", 0.0) / wcyc, "wait_inst_any_frac": s.get("SQ_WAIT_INST_ANY", 0.0) / wcyc,
                   "active_inst_any_frac": s.get("SQ_ACTIVE_INST_ANY", 0.0) / wcyc,
                   "waves_per_launch": s.get("SQ_WAVES", 0.0) / max(sc[fam].get("SQ_WAVES", 1), 1),
                   "mfma_insts_per_launch": s.get("SQ_INSTS_MFMA", 0.0) / max(sc[fam].get("SQ_INSTS_MFMA", 1), 1),
                   "valu_insts_per_launch": s.get("SQ_INSTS_VALU", 0.0) / max(sc[fam].get("SQ_INSTS_VALU", 1), 1)}
    l = la.get(fam)
    if l and l.get("SQ_INSTS_LDS"):
        e["lds"] = {"bank_conflict_cycles_per_lds_inst": l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_INSTS_LDS"],
                    "lds_insts_per_launch": l["SQ_INSTS_LDS"] / max(lc[fam].get("SQ_INSTS_LDS", 1), 1)}
    fams[fam] = e
# algorithmic bytes per launch of the SAME run (what bench.py's rooflines count), from the counters scripts/probe_batch.py prints
import re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cnt = None
for name in ("fetch", "write"):
    try:
        mo = re.search(r"COUNTERS blocks_nonempty=(\d+) updates_accepted=(\d+) qr_calls=(\d+) chains=(\d+) sweeps_total=(\d+) n_g=(\d+)",
                       open(os.path.join(src, name + ".log")).read())
        if mo:
            cnt = [int(x) for x in mo.groups()]
            break
    except OSError:
        pass
if cnt:
    from bench import qr_apply_work
    blocks, acc, qr_calls, chains, _, n = cnt
    MSF = 2
    alg = {"bmult": 2 * 16.0 * n * n * chains, "gemm": 3 * 16.0 * n * n * chains,
           "flush": (2 * 16.0 * n * n * blocks + 2 * 16.0 * n * MSF * acc) / max(fams.get("flush", {}).get("dispatches", 1), 1),
           "gather": 4 * 16.0 * n * MSF * acc / max(fams.get("gather", {}).get("dispatches", 1), 1),
           "qr_apply": qr_apply_work(n)[0] * qr_calls * chains / max(fams.get("qr_apply", {}).get("dispatches", 1), 1)}
    for fam, a in alg.items():
        if fam in fams:
            fams[fam]["algorithmic_bytes_per_launch_same_run"] = a
            fams[fam]["traffic_over_algorithmic"] = fams[fam]["hbm_bytes_per_launch"] / a
doc = {"workload": "DetSDW O(2) L=16 beta=10 (n_g=512), ONE context of %d lockstep chains, delaySteps %d, scripts/probe_batch.py 16 10 1 qr %d "
                   "(2 warm-up sweeps + 1 timed sweep, every dispatch counted)" % (B, D, B),
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)",
       "families": fams}
json.dump(doc, open(out, "w"), indent=1)
for k, v in fams.items():
    print("%-10s %6d dispatches  %8.1f MB/launch (read %.1f, write %.1f)%s%s" % (k, v["dispatches"], v["hbm_bytes_per_launch"] / 1e6, v["read_bytes_per_launch"] / 1e6,
          v["write_bytes_per_launch"] / 1e6, ("  x%.2f of algorithmic" % v["traffic_over_algorithmic"]) if "traffic_over_algorithmic" in v else "", "  wait_any %.2f active %.2f" % (v["sq"]["wait_any_frac"], v["sq"]["active_inst_any_frac"]) if "sq" in v else ""))
