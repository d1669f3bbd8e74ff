"""numpy twin of the native batch packer (csrc/pf_pack.cpp -> packing.build_batch_native): the same arrays built
record by record in Python.  TEST INFRASTRUCTURE: the product packs through the library; tests use this twin to check
the native packer array for array and to pack records for device-resident batches."""
import numpy as np

from panfeed_amd.packing import _COMP_CODE, _LUT, HostBatch, SeqMeta, pack_codes


def build_batch(records, klength, canon, W, stroi=(), first_ordinal=0, want_strand=True):
    """records: iterable of (gene_sequences, idx, clusterpresab)."""
    k = int(klength)
    hb = HostBatch(k=k, canon=bool(canon), W=W)
    words, word_off = [], 0
    seg_word_off, seg_len, seg_sample, seg_ord, seg_strand = [], [], [], [], []
    cl_seg_off = [0]
    cl_nstr, cl_npres, cl_presab, cl_ord = [], [], [], []
    ex_cluster, ex_ord, ex_bits = [], [], []
    strand_words = 0

    for ci, (gs, idx, presab) in enumerate(records):
        names = list(gs.keys())
        n = len(names)
        if n > W * 32:
            raise ValueError(f"cluster {idx}: {n} strains exceed the context's max_strains")
        sorted_names = sorted(names)
        col = {x: i for i, x in enumerate(sorted_names)}          # panfeed.py:47-49
        presab = np.asarray(presab)
        if len(presab) > W * 32:
            raise ValueError(f"cluster {idx}: clusterpresab longer than max_strains")
        if presab.size and not np.isin(presab, (0, 1)).all():
            raise ValueError(f"cluster {idx}: clusterpresab must hold 0/1")
        pb = np.zeros(W, dtype=np.uint32)
        for i in np.flatnonzero(presab):
            pb[i >> 5] |= np.uint32(1 << (i & 31))
        hb.idx.append(str(idx))
        hb.sorted_strains.append(sorted_names)
        hb.presab.append(presab)
        cl_nstr.append(n)
        cl_npres.append(len(presab))
        cl_presab.append(pb)
        cl_ord.append(first_ordinal + ci)

        segs = []          # (sample, order, word arrays, len, ord_base, strand_off)
        ambig_rows = {}    # key -> [first_ord, set(cols)]
        ord_base = 0
        order = 0
        for strain in names:                                      # panfeed.py:54
            c = col[strain]
            is_target = bool(stroi) and (strain in stroi)         # panfeed.py:90
            for s in gs[strain]:                                  # panfeed.py:55
                seq = s.sequence
                L = len(seq)
                num_kmer = max(L - k + 1, 0)                      # panfeed.py:59,64
                raw = np.frombuffer(seq.encode("latin-1"), dtype=np.uint8)
                codes = _LUT[raw]
                bad = np.flatnonzero(codes == 255)
                craw = np.frombuffer(s.compsequence.encode("latin-1"), dtype=np.uint8)
                if len(craw) != L:
                    raise ValueError(f"{idx}/{strain}: sequence and compsequence differ in length")
                ok = codes != 255
                if not np.array_equal(_LUT[craw][ok], _COMP_CODE[codes[ok]]):
                    raise ValueError(f"{idx}/{strain}: compsequence is not the complement of sequence")
                meta = SeqMeta(ci, strain, s, ord_base, num_kmer, [], {}) if is_target else None
                # maximal A/C/G/T runs -> device segments
                bounds = np.concatenate(([-1], bad, [L]))
                for a, b in zip(bounds[:-1] + 1, bounds[1:]):
                    a, b = int(a), int(b)
                    if b - a < k:
                        continue
                    soff = 0xFFFFFFFF
                    if is_target and canon and want_strand:
                        soff = strand_words
                        strand_words += (b - a - k + 1 + 63) // 64
                    segs.append((c, order, pack_codes(codes[a:b]), b - a, ord_base + a, soff))
                    if meta is not None:
                        meta.segs.append((order, a, b - a - k + 1))
                    order += 1
                # windows touching a non-ACGT base: the reference's own string semantics
                if len(bad) and num_kmer > 0:
                    comp = s.compsequence
                    touched = np.zeros(num_kmer, dtype=bool)
                    for p in bad:
                        touched[max(0, int(p) - k + 1):min(num_kmer, int(p) + 1)] = True
                    for pos in np.flatnonzero(touched):
                        pos = int(pos)
                        spec = seq[pos:pos + k]                   # panfeed.py:65
                        rev = comp[pos:pos + k][::-1]             # panfeed.py:67
                        if canon:
                            key, used = (spec, 1) if spec <= rev else (rev, -1)   # panfeed.py:70-75
                            ent = ambig_rows.setdefault(key, [ord_base + pos, set()])
                            ent[1].add(c)
                            if meta is not None:
                                meta.ambig[pos] = (key, used)
                        else:
                            for j, key in enumerate((spec, rev)):                 # panfeed.py:82-88
                                ent = ambig_rows.setdefault(key, [2 * (ord_base + pos) + j, set()])
                                ent[1].add(c)
                if meta is not None:
                    hb.targets.append(meta)
                ord_base += num_kmer
        hb.n_instances += ord_base * (1 if canon else 2)
        if ord_base * 2 >= 0xFFFFFFF0:
            raise ValueError(f"cluster {idx}: too many k-mer instances for 32-bit ordinals")

        # segments sorted by sample column (stable); remember where each went for the strand bits
        segs_sorted = sorted(segs, key=lambda t: (t[0], t[1]))
        base_index = len(seg_len)
        where = {}
        for j, (c, order, w, ln, ob, soff) in enumerate(segs_sorted):
            where[order] = base_index + j
            seg_word_off.append(word_off)
            words.append(w)
            word_off += len(w)
            seg_len.append(ln)
            seg_sample.append(c)
            seg_ord.append(ob)
            seg_strand.append(soff)
        for meta in hb.targets:
            if meta.cluster == ci:
                meta.segs = [(where[o], a, nw) for (o, a, nw) in meta.segs]
        cl_seg_off.append(len(seg_len))
        for key, (o, cols) in ambig_rows.items():
            row = np.zeros(W, dtype=np.uint32)
            for c in cols:
                row[c >> 5] |= np.uint32(1 << (c & 31))
            ex_cluster.append(ci)
            ex_ord.append(o)
            ex_bits.append(row)
            hb.extra_keys.append(key)

    words.append(np.zeros(4, dtype=np.uint64))   # 32 bytes of tail padding
    hb.packed = np.ascontiguousarray(np.concatenate(words))
    hb.seg_word_off = np.asarray(seg_word_off, dtype=np.uint64)
    hb.seg_len = np.asarray(seg_len, dtype=np.uint32)
    hb.seg_sample = np.asarray(seg_sample, dtype=np.uint32)
    hb.seg_ord_base = np.asarray(seg_ord, dtype=np.uint32)
    hb.seg_strand_off = np.asarray(seg_strand, dtype=np.uint32)
    hb.n_strand_words = strand_words
    hb.cluster_seg_off = np.asarray(cl_seg_off, dtype=np.uint32)
    hb.cluster_nstrains = np.asarray(cl_nstr, dtype=np.uint32)
    hb.cluster_npresab = np.asarray(cl_npres, dtype=np.uint32)
    hb.cluster_presab = (np.stack(cl_presab) if cl_presab else np.zeros((0, W), dtype=np.uint32)).astype(np.uint32)
    hb.cluster_ordinal = np.asarray(cl_ord, dtype=np.uint64)
    hb.extra_cluster = np.asarray(ex_cluster, dtype=np.uint32)
    hb.extra_ord = np.asarray(ex_ord, dtype=np.uint32)
    hb.extra_bits = (np.stack(ex_bits) if ex_bits else np.zeros((0, W), dtype=np.uint32)).astype(np.uint32)
    return hb
