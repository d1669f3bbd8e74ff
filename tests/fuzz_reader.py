"""Randomised inputs for the native reader (not collected by pytest; tools/asan_host.sh runs it against a build whose host
code carries AddressSanitizer / UBSan): per case a random small pangenome, every genome's FASTA re-wrapped in a style of its
own (tests/test_native_input.py: _mangle_fasta), random flanks and k -- the one-pass reader, the text-mode reader and the
Python restatement of the reference's reader must describe the same input.

    python tests/fuzz_reader.py [cases [first_seed]]
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]


def main():
    import pathlib

    import test_native_input as t
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tot = np.zeros(3, dtype=np.int64)
    for seed in range(first, first + cases):
        rng = np.random.default_rng(77000 + seed)
        styles = {}

        def style_of(n, styles=styles, rng=rng):
            if n not in styles:
                styles[n] = t._STYLES[int(rng.integers(0, len(t._STYLES)))]
            return styles[n]
        kw = dict(n_clusters=int(rng.integers(1, 20)), n_samples=int(rng.integers(2, 40)), first=int(rng.integers(0, 10**5)),
                  mean_len=int(rng.choice([40, 260, 900])), min_len=int(rng.choice([3, 40])), max_len=1500,
                  n_rate=float(rng.choice([0.0, 0.08, 0.5])), paralog_rate=float(rng.choice([0.0, 0.1, 0.3])))
        with tempfile.TemporaryDirectory() as d:
            tot += t.check_one_pass(pathlib.Path(d), style_of, mangle_seed=seed, synth_kw=kw, up=int(rng.choice([0, 30, 400, 5000])),
                                    down=int(rng.choice([0, 20, 300, 5000])), k=int(rng.choice([5, 21, 31, 63, 95])),
                                    max_clusters=int(rng.integers(1, 8)), strict=False)
        with tempfile.TemporaryDirectory() as d:
            t.test_reader_fuzz_against_restatement(pathlib.Path(d), 5000 + seed)
        if (seed - first + 1) % 20 == 0:
            print(f"{seed - first + 1} cases ok", flush=True)
    print(f"{cases} cases: {int(tot[0])} segments by reference ({int(tot[1])} reverse strand), {int(tot[2])} literal -- all equal")


if __name__ == "__main__":
    main()
