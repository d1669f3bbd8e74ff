"""Randomised differential test on the GPU: random options and cluster shapes, HIP path vs the oracle, text for text.
usage: python tests/fuzz_parity.py [n_cases] [seed] [big]   (test infrastructure: imports oracle/)"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as po
from panfeed_amd import synth
from panfeed_amd.engine import Engine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
big = len(sys.argv) > 3 and sys.argv[3] == "big"        # BASELINE-sized clusters: hundreds to 1 000 samples
fails = 0
verbose = bool(os.environ.get("PF_FUZZ_VERBOSE"))
seen = {"binned": 0, "wide": 0, "retried": 0, "partitioned": 0}      # cases that took the path at least once
t0 = time.time()
only = os.environ.get("PF_FUZZ_ONLY")                     # one case of the run, by its number
for case in range(n_cases):
    if only is not None and case != int(only):
        continue
    rng = np.random.default_rng(seed0 * 100003 + case)
    k = int(rng.choice([3, 7, 15, 21, 31, 32, 33, 47, 63, 64, 80, 95, 126]))
    S = int(rng.choice([300, 640, 1000, 1100])) if big else int(rng.choice([5, 17, 33, 64, 90, 130, 260]))
    ncl = int(rng.integers(1, 5 if big else 9))
    kw = dict(klength=k, canon=bool(rng.random() < 0.7), consider_missing=bool(rng.random() < 0.3),
              patfilt=bool(rng.random() < 0.8), maf=float(rng.choice([0.0, 0.01, 0.05, 0.2])))
    gen = dict(flank=int(rng.choice([0, 0, 10, 60])), mean_len=int(rng.choice([60, 150, 400, 1500])), min_len=int(rng.choice([5, 40])),
               max_len=int(rng.choice([300, 2500])), n_rate=float(rng.choice([0.0, 0.02, 0.2])),
               paralog_rate=float(rng.choice([0.0, 0.05, 0.3])), sub_rate=float(rng.choice([0.0, 0.01, 0.1])),
               mean_alleles=float(rng.choice([1.0, 7.0, 40.0, 120.0] if big else [1.0, 7.0, 40.0])),
               allele_decay=float(rng.choice([0.5, 1.0])), allele_model=str(rng.choice(["star", "tree"])))
    if gen["min_len"] > gen["max_len"]:
        gen["min_len"] = gen["max_len"]
    shuffle = int(rng.integers(0, 99)) if rng.random() < 0.5 else None
    cl = synth.generate(ncl, S, first=int(rng.integers(0, 10**6)), shuffle_columns=shuffle, **gen)
    recs = [c.record() for c in cl]
    names = cl[0].names
    stroi = set(rng.choice(names, size=min(len(names), int(rng.integers(0, 3))), replace=False).tolist())
    dedup = bool(rng.random() < 0.8)
    unit_dedup = bool(rng.random() < 0.85)
    key_binning = bool(rng.random() < 0.8)
    device_plan = bool(rng.random() < 0.5)
    max_items = int(rng.choice([64, 2048]))
    cut = int(rng.integers(0, ncl + 1))
    if verbose:        # (PF_FUZZ_VERBOSE=1: what is about to run, for a case that never comes back)
        print("case", case, kw, gen, "S", S, "ncl", ncl, "dedup", dedup, "stroi", len(stroi), "shuffle", shuffle, "unit_dedup", unit_dedup,
              "key_binning", key_binning, "device_plan", device_plan, "max_items", max_items, "cut", cut, flush=True)
    try:
        for attempt in range(2):
            eng = Engine(max_strains=(S + 31) // 32 * 32, stroi=stroi, dedup=dedup, unit_dedup=unit_dedup, key_binning=key_binning, device_plan=device_plan, max_items=max_items, **kw)
            try:
                outs, hit = [], set()
                for part in ([recs[:cut]] if cut else []) + ([recs[cut:]] if cut < ncl else []):
                    outs.append(eng.run(part))
                    t = eng.timing()
                    hit |= {n for n, f in (("binned", "n_binned_clusters"), ("wide", "n_wide_clusters"), ("retried", "n_retried")) if t[f]}
                    if t["n_items"] > len(part):
                        hit.add("partitioned")
                for n in hit:
                    seen[n] += 1
                break
            except Exception as e:  # noqa: BLE001
                # (a cluster that needs more work items than this tiny setting allows makes the context re-make its scratch
                # since round 4; the retry stays for libraries older than that)
                if attempt == 0 and "raise max_items" in repr(e):
                    max_items = 8192
                    continue
                raise
            finally:
                eng.close()
        run = po.OracleRun(stroi=stroi, threads=16, **kw)
        run.feed(recs)
        ek, ekh, ehp = run.texts()
        got = ("".join(o.kmers_tsv for o in outs), "".join(o.kmers_to_hashes for o in outs), "".join(o.hashes_to_patterns for o in outs))
        ok = got == (ek, ekh, ehp)
    except Exception as e:  # noqa: BLE001
        ok = False
        print("EXC", repr(e)[:200])
    if not ok:
        fails += 1
        print("FAIL case", case, kw, gen, "S", S, "ncl", ncl, "dedup", dedup, "stroi", len(stroi), "shuffle", shuffle, "unit_dedup", unit_dedup, "key_binning", key_binning, "device_plan", device_plan,
              "max_items", max_items, "cut", cut, flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print("DONE", n_cases, "cases", fails, "failures; cases by path:", seen)
sys.exit(1 if fails else 0)
