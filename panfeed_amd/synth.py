"""Seeded synthetic gene clusters (the shapes BASELINE.json's configs name).

No reference code is involved: this produces the *input* of the hot path, either as
reference-shaped records ``(gene_sequences, idx, clusterpresab)`` (the tuple
/root/reference/panfeed/input.py:468 yields) for parity tests, or as allele pools +
per-sequence allele ids for the device-side expansion bench.py uses at full size.

Per cluster c: rng = PCG64(0x9E3779B9 ^ c); gene length ~ lognormal(ln mean_len, 0.5)
clipped to [min_len, max_len]; 1 + Poisson(7) alleles = ancestral uniform-ACGT sequence
with allele-specific 1 % substitutions and allele-specific random flanks; allele weights
~ 2^-i (`allele_decay`^i: 1.0 spreads the samples evenly over the alleles, which is how the bench sweeps the number
of distinct sequences per cluster); cluster frequency 0.99 (60 % "core") or U(0.02, 0.9); 1 % of present samples
carry a paralog copy; strand +-1 (affects coordinates only); `n_rate` of the sequences
get one 'N'.

`allele_model="tree"` (not SURVEY's; bench.py's second allele sweep): the alleles of a cluster descend from one
another the way a population's do -- allele i is a copy of a random earlier allele, flanks included, with
`tree_mutations` point substitutions -- so that many distinct sequences share most of their k-mers.
"""
from dataclasses import dataclass, field

import numpy as np

from .classes import Seqinfo

_CODE2ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = bytes.maketrans(b"ACGTN", b"TGCAN")


@dataclass
class SynthCluster:
    index: int                 # global cluster ordinal (processing order)
    idx: str                   # cluster name
    names: list                # all sample names, CSV column order
    alleles: list              # list of uint8 code arrays (0..3), flanks included
    up: int                    # upstream flank actually included (Seqinfo.offset)
    seq_sample: np.ndarray     # per sequence (iteration order): index into names
    seq_allele: np.ndarray
    seq_strand: np.ndarray
    seq_npos: np.ndarray       # position of a single 'N', or -1
    present: np.ndarray        # bool[S], CSV column order
    _strs: dict = field(default_factory=dict, repr=False)

    @property
    def n_seqs(self):
        return len(self.seq_sample)

    def n_instances(self, k, canon=True):
        lens = np.array([len(a) for a in self.alleles], dtype=np.int64)[self.seq_allele]
        n = np.maximum(lens - k + 1, 0).sum()
        return int(n if canon else 2 * n)

    def seq_string(self, q):
        a = int(self.seq_allele[q])
        s = self._strs.get(a)
        if s is None:
            s = _CODE2ASCII[self.alleles[a]].tobytes()
            self._strs[a] = s
        p = int(self.seq_npos[q])
        if p >= 0:
            s = s[:p] + b"N" + s[p + 1:]
        return s.decode()

    def record(self):
        """(gene_sequences, idx, clusterpresab) exactly as input.py:366-468 lays it out."""
        names = self.names
        sorted_names = sorted(names)
        col = {x: i for i, x in enumerate(sorted_names)}
        presab = np.zeros(len(names), dtype=np.int64)
        gs = {}
        for q in range(self.n_seqs):
            name = names[int(self.seq_sample[q])]
            seq = self.seq_string(q)
            comp = seq.encode().translate(_COMP).decode()
            lst = gs.setdefault(name, [])
            strand = int(self.seq_strand[q])
            start = 1000 + 37 * q
            lst.append(Seqinfo(seq, comp, f"{name}_{self.index:05d}_{len(lst)}", f"{name}_contig1",
                               start, start + len(seq) - 1, strand, self.up))
        for i, name in enumerate(names):
            if self.present[i]:
                presab[col[name]] = 1
                gs.setdefault(name, [])
        # present strains first (CSV order) -- a dict keeps first-insertion order, so re-create
        ordered = {}
        for i, name in enumerate(names):
            if self.present[i]:
                ordered[name] = gs[name]
        for name in sorted_names:      # `absent` is a pandas Index.difference -> sorted (input.py:373,465)
            if name not in ordered:
                ordered[name] = []
        return ordered, self.idx, presab


def sample_names(n, shuffle_seed=None):
    names = [f"s{i:05d}" for i in range(n)]
    if shuffle_seed is not None:
        np.random.Generator(np.random.PCG64(shuffle_seed)).shuffle(names)
    return names


def generate_cluster(c, names, flank=0, mean_len=900, min_len=150, max_len=6000, n_rate=0.001,
                     paralog_rate=0.01, sub_rate=0.01, mean_alleles=7.0, seed=0x9E3779B9, allele_decay=0.5,
                     allele_model="star", tree_mutations=2):
    rng = np.random.Generator(np.random.PCG64(seed ^ c))
    S = len(names)
    L = int(np.clip(np.rint(rng.lognormal(np.log(mean_len), 0.5)), min_len, max_len))
    H = 1 + int(rng.poisson(mean_alleles))
    anc = rng.integers(0, 4, size=L, dtype=np.uint8)
    alleles = []
    if allele_model == "tree":
        root = anc if not flank else np.concatenate([rng.integers(0, 4, size=flank, dtype=np.uint8), anc,
                                                     rng.integers(0, 4, size=flank, dtype=np.uint8)])
        alleles.append(root)
        seen = {root.tobytes()}
        while len(alleles) < H:
            a = alleles[int(rng.integers(0, len(alleles)))].copy()
            at = rng.integers(0, len(a), size=tree_mutations)
            a[at] = (a[at] + rng.integers(1, 4, size=tree_mutations, dtype=np.uint8)) & 3
            if a.tobytes() not in seen:
                seen.add(a.tobytes())
                alleles.append(a)
    elif allele_model != "star":
        raise ValueError(f"allele_model {allele_model!r}")
    for i in range(H if allele_model == "star" else 0):
        a = anc.copy()
        if i > 0:
            mut = rng.random(L) < sub_rate
            a[mut] = (a[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
        if flank:
            a = np.concatenate([rng.integers(0, 4, size=flank, dtype=np.uint8), a,
                                rng.integers(0, 4, size=flank, dtype=np.uint8)])
        alleles.append(a)
    w = float(allele_decay) ** np.arange(H)     # 0.5: SURVEY 8d's 2^-i; 1.0: every allele equally common
    w /= w.sum()
    p = 0.99 if rng.random() < 0.6 else rng.uniform(0.02, 0.9)
    present = rng.random(S) < p
    if not present.any():
        present[int(rng.integers(0, S))] = True
    pres_idx = np.flatnonzero(present)
    copies = 1 + (rng.random(len(pres_idx)) < paralog_rate).astype(np.int64)
    seq_sample = np.repeat(pres_idx, copies).astype(np.int32)
    nseq = len(seq_sample)
    seq_allele = rng.choice(H, size=nseq, p=w).astype(np.int32)
    seq_strand = np.where(rng.random(nseq) < 0.5, 1, -1).astype(np.int8)
    seq_npos = np.full(nseq, -1, dtype=np.int32)
    has_n = rng.random(nseq) < n_rate
    for q in np.flatnonzero(has_n):
        seq_npos[q] = int(rng.integers(0, len(alleles[seq_allele[q]])))
    return SynthCluster(c, f"group_{c:06d}", list(names), alleles, flank, seq_sample, seq_allele,
                        seq_strand, seq_npos, present)


def generate(n_clusters, n_samples, first=0, shuffle_columns=None, **kw):
    names = sample_names(n_samples, shuffle_columns)
    return [generate_cluster(first + i, names, **kw) for i in range(n_clusters)]


def _sample_files(outdir, nm, genes, rng, wrap, missing_gene_rate, lower_rate, drop, separate):
    """one sample's GFF3 (+ FASTA): (gff path or None, fasta path or None)"""
    import os
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    ncontig = int(rng.integers(1, 4))
    contigs = [[] for _ in range(ncontig)]
    for g in genes:
        contigs[int(rng.integers(0, ncontig))].append(g)
    gff_lines = ["##gff-version 3"]
    fasta_lines = []
    for c, glist in enumerate(contigs):
        cname = f"{nm}_contig{c + 1}"
        seq = bytearray()
        first = True
        for gid, s, strand, inset in glist:
            spacer = 0 if (first and rng.random() < 0.3) else int(rng.integers(3, 180))
            first = False
            seq += np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, spacer)].tobytes()
            start = len(seq) + 1 + inset          # clusters made with flank=F: the CDS is the inner part, so
            seq += s if strand > 0 else s[::-1].translate(comp)
            end = len(seq) - inset                # reading with upstream=downstream=F returns the whole allele
            if rng.random() >= missing_gene_rate:
                gff_lines.append(f"{cname}\tProdigal\tCDS\t{start}\t{end}\t.\t{'+' if strand > 0 else '-'}\t0\t"
                                 f"ID={gid};Parent={gid}_gene;product=hypothetical protein")
                gff_lines.append(f"{cname}\tProdigal\tgene\t{start}\t{end}\t.\t{'+' if strand > 0 else '-'}\t.\tID={gid}_gene")
        if rng.random() < 0.7:
            seq += np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(rng.integers(1, 150)))].tobytes()
        seq = bytes(seq)
        if lower_rate:
            arr = np.frombuffer(seq, np.uint8).copy()
            low = rng.random(len(arr)) < lower_rate
            arr[low] |= 0x20
            seq = arr.tobytes()
        fasta_lines.append(f">{cname} len={len(seq)}")
        fasta_lines += [seq[i:i + wrap].decode() for i in range(0, len(seq), wrap)]
        gff_lines.insert(1 + c, f"##sequence-region {cname} 1 {len(seq)}")
    if drop:
        return None, None
    path = os.path.join(outdir, "gffs", f"{nm}.gff")
    fa = None
    with open(path, "w") as fh:
        fh.write("\n".join(gff_lines) + "\n")
        if separate:
            fa = os.path.join(outdir, "gffs", f"{nm}.fasta")
            with open(fa, "w") as f2:
                f2.write("\n".join(fasta_lines) + "\n")
        else:
            fh.write("##FASTA\n" + "\n".join(fasta_lines) + "\n")
    return path, fa


_POOL_ARGS = None      # (clusters, outdir, seed, wrap, ...) of a parallel write_pangenome, inherited by its forked workers


def _pool_sample(si):
    clusters, outdir, seed, wrap, missing_gene_rate, lower_rate, drop_gff_for, separate_fasta_for = _POOL_ARGS
    nm = clusters[0].names[si]
    genes = []
    for cl in clusters:
        k = 0
        for q in np.flatnonzero(cl.seq_sample == si):
            genes.append((f"{nm}_{cl.index:05d}_{k}", cl.seq_string(int(q)).encode(), int(cl.seq_strand[q]), int(cl.up)))
            k += 1
    rng = np.random.Generator(np.random.PCG64([seed, si]))
    return nm, _sample_files(outdir, nm, genes, rng, wrap, missing_gene_rate, lower_rate, nm in drop_gff_for, nm in separate_fasta_for)


def write_pangenome(outdir, clusters, seed=1, wrap=60, drop_gff_for=(), missing_gene_rate=0.01, lower_rate=0.02,
                    separate_fasta_for=(), workers=0):
    """Write `clusters` as an on-disk pangenome the way panfeed reads it: one Prokka-style GFF3 per sample
    (CDS features, ##FASTA section) and a panaroo gene_presence_absence.csv.  Genes sit on 1-3 contigs per sample with
    random spacers, some flush against a contig edge (offset clipping), - strand genes stored reverse-complemented.
    workers > 0: the samples' files are written by that many forked processes, each sample from a random stream of its
    own (another layout than workers=0 gives for the same seed -- both are valid pangenomes of the same clusters).
    Returns (csv_path, {sample: gff_path}, {sample: fasta_path or None})."""
    import os
    names = clusters[0].names
    cells = [dict() for _ in clusters]
    os.makedirs(os.path.join(outdir, "gffs"), exist_ok=True)
    gffs, fastas = {}, {}
    if workers > 0:
        import multiprocessing as mp
        global _POOL_ARGS
        for ci, cl in enumerate(clusters):
            copies = {}
            for q in range(cl.n_seqs):
                nm = names[int(cl.seq_sample[q])]
                k = copies.get(nm, 0)
                copies[nm] = k + 1
                cells[ci].setdefault(nm, []).append(f"{nm}_{cl.index:05d}_{k}")
        _POOL_ARGS = (clusters, outdir, seed, wrap, missing_gene_rate, lower_rate, set(drop_gff_for), set(separate_fasta_for))
        try:
            with mp.get_context("fork").Pool(workers) as pool:
                for nm, (path, fa) in pool.imap_unordered(_pool_sample, range(len(names)), chunksize=4):
                    if path is not None:
                        gffs[nm] = path
                        fastas[nm] = fa
        finally:
            _POOL_ARGS = None
    else:
        rng = np.random.Generator(np.random.PCG64(seed))
        per_sample = {nm: [] for nm in names}          # (gene id, seq bytes, strand, inset)
        for ci, cl in enumerate(clusters):
            copies = {}
            for q in range(cl.n_seqs):
                nm = names[int(cl.seq_sample[q])]
                k = copies.get(nm, 0)
                copies[nm] = k + 1
                gid = f"{nm}_{cl.index:05d}_{k}"
                per_sample[nm].append((gid, cl.seq_string(q).encode(), int(cl.seq_strand[q]), int(cl.up)))
                cells[ci].setdefault(nm, []).append(gid)
        for nm in names:
            path, fa = _sample_files(outdir, nm, per_sample[nm], rng, wrap, missing_gene_rate, lower_rate,
                                     nm in drop_gff_for, nm in separate_fasta_for)
            if path is not None:
                gffs[nm] = path
                fastas[nm] = fa
    csv_path = os.path.join(outdir, "gene_presence_absence.csv")
    with open(csv_path, "w") as fh:
        fh.write(",".join(["Gene", "Non-unique Gene name", "Annotation"] + names) + "\n")
        for ci, cl in enumerate(clusters):
            row = [cl.idx, "", '"hypothetical protein, putative"'] + [";".join(cells[ci].get(nm, [])) for nm in names]
            fh.write(",".join(row) + "\n")
    return csv_path, gffs, fastas
