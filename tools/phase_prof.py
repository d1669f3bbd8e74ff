"""Cycles per phase inside kmer_scan_kernel, rows_kernel and emit_kernel (PF_PROF build; finish_kernel's stamps are off unless
-DPF_PROF_FINISH is given too -- with them it faults on batches of thousands of clusters).  Here:
    bash tools/ab_bench.sh build prof "-DPF_PROF"
and on the GPU box, with ab/prof.so copied over panfeed_amd/libpanfeed_hip.so for the run:
    python tools/phase_prof.py [clusters [mean_alleles star|tree]]"""
import ctypes as C
import sys

sys.path.insert(0, ".")
from panfeed_amd import devbatch, synth  # noqa: E402
from panfeed_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
S, k = 1000, 31
ma = float(sys.argv[2]) if len(sys.argv) > 2 else None
eng = Engine(klength=k, max_strains=1024, max_items=32768 if ma else 8192, pattern_capacity=1 << 24 if ma else 1 << 23)
if ma:
    cl = synth.generate(n, S, flank=100, n_rate=0.0, mean_alleles=ma, allele_decay=1.0, allele_model=sys.argv[3])
else:
    cl = synth.generate(n, S, flank=100, n_rate=0.0)
db = devbatch.from_synth(eng, cl, k)
db.submit()
eng.L.pf_reset_patterns(eng.ctx)
buf = (C.c_uint64 * 64)()
eng.L.pf_debug_prof(buf, 1)
db.submit()
eng.L.pf_debug_prof(buf, 0)
v = list(buf)
names = {0: "finish: init + M", 1: "finish: A masks", 2: "finish: B row eval", 3: "finish: C bitmaps", 4: "finish: prefix",
         5: "finish: find/claim", 6: "finish: publish+rows", 7: "finish: outputs", 8: "finish: workgroups",
         16: "scan: clear + tile0", 17: "scan: unit prefix", 18: "scan: windows", 19: "scan: next desc/tile",
         20: "scan: table dump", 21: "scan: desc swap", 24: "scan: items",
         32: "emit: init + cluster row", 33: "emit: pass 1 (rank, local table)", 34: "emit: pass 2 (global inserts)",
         35: "emit: pass 3 (outputs)", 36: "emit: items",
         40: "rows(wide): segments", 41: "rows(wide): A mask table", 42: "rows(wide): C rows", 43: "rows(wide): C hashes + D",
         44: "rows: slots", 45: "rows: bitmaps out", 46: "rows: sort + out", 47: "rows(wide): items",
         48: "rows(wide): distinct masks", 49: "rows(wide): slots", 50: "rows(wide): rounds"}
for lo, hi, cnt in ((0, 8, 8), (16, 24, 24), (32, 36, 36), (40, 47, 47)):
    tot = sum(v[lo:hi])
    for i in list(range(lo, hi)) + [cnt]:
        if v[i]:
            print(f"{i:2d} {names.get(i, ''):24s} {v[i]:14d} {100.0 * v[i] / tot if i < hi else 0:6.1f}%"
                  + (f"  {v[i] / v[cnt]:9.0f} cyc/item" if i < hi and v[cnt] else ""))
if v[36]:
    print(f"37 emit: distinct patterns in the local table {v[37] / v[36]:9.1f} per item")
for i in (48, 49, 50):
    if v[47]:
        print(f"{i:2d} {names[i]:28s} {v[i] / v[47]:9.1f} per item")
if v[47]:
    print(f"step A (wave 0): clear {v[55] / v[47]:.0f}, own slots {v[56] / v[47]:.0f}, waiting for the others {v[41] / v[47]:.0f} cycles per item")
if v[38]:
    print(f"emit pass 1, wave 0: {v[38] / v[36]:.1f} trips per item; per trip {v[62] / v[38]:.0f} cycles until the loads are in, {v[63] / v[38]:.0f} in the local table")
if v[60]:
    print(f"rows, mode-1 items ({v[60]}): M {v[57] / v[60]:.0f}, mask table {v[58] / v[60]:.0f}, rows of {v[61] / v[60]:.0f} masks {v[59] / v[60]:.0f}, "
          f"slot loop {(v[44] - (v[44] if not v[47] else 0)) / max(v[60], 1):.0f} (with the wide items' share) cycles per item")
print(eng.timing())
