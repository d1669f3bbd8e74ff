"""Python restatement of the serial producer in front of the hot path.  TEST INFRASTRUCTURE ONLY.

    parse_gff            /root/reference/panfeed/input.py:274-332
    iter_gene_clusters   /root/reference/panfeed/input.py:335-468
    panaroo table load   /root/reference/panfeed/input.py:188-191

PINNED by tests/golden/n1.json.gz (tools/gen_golden_n1.py: the reference's own `parse_gff`, `prep_data_n_fasta`,
`set_input_output` and `iter_gene_clusters` run in the build container; tests/test_native_input.py::
test_restatement_equals_reference_goldens) -- EXCEPT for pyfaidx, which is neither in /root/reference nor installed:
the goldens were made with a declared double of `pyfaidx.Fasta` (record key = header up to the first whitespace;
sequence without line breaks; sequence_always_upper=True; Python-slice clipping; `-seq` = reverse complement with the
table below; `[::-1]`; `str`).  PARITY UNPINNED for those five operations, pinned for everything around them.
This file checks the native reader (panfeed_amd/csrc/pf_input.cpp) on inputs the goldens do not cover.
"""
import csv

import numpy as np

from panfeed_amd.classes import Feature, Seqinfo

_COMP = str.maketrans("ACTGNactgnYRWSKMDVHBXyrwskmdvhbx", "TGACNtgacnRYWSMKHBDVXryswmkhbdvx")
_NA = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A",
       "NA", "NULL", "NaN", "None", "n/a", "nan", "null"}


def read_fasta(text):
    out, name, chunks = {}, None, []
    for line in text.split("\n"):
        line = line.rstrip("\r")
        if line.startswith(">"):
            if name is not None and name not in out:
                out[name] = "".join(chunks).upper()
            name, chunks = line[1:].split()[0] if line[1:].split() else "", []
        elif name is not None:
            chunks.append(line)
    if name is not None and name not in out:
        out[name] = "".join(chunks).upper()
    return out


def parse_gff(file_name):
    features, warnings = {}, []
    with open(file_name) as gff:
        for line in gff:
            if line.lstrip().startswith("##FASTA"):
                break
            elif line.lstrip().startswith("#"):
                continue
            entries = line.split("\t")
            try:
                ftype = entries[2]
                if ftype not in {"CDS"}:
                    continue
                chrom = entries[0]
                start = int(entries[3])
                end = int(entries[4])
                strand = 1 if entries[6] == "+" else -1
                ID = None
                for entry in entries[8].split(";"):
                    if entry.startswith("ID") and "=" in entry:
                        ID = entry.split("=")[1]
                if ID is None:
                    continue
                features[ID] = Feature(ID, chrom, start, end, strand)
            except Exception as e:
                warnings.append(f'{e}, skipping line "{line.rstrip()}" from {file_name}')
                continue
    return features, warnings


def load_table(path):
    with open(path, newline="") as fh:
        rows = list(csv.reader(fh))
    header = rows[0]
    keep = [i for i, h in enumerate(header) if i > 0 and h not in ("Non-unique Gene name", "Annotation")]
    strains = [header[i] for i in keep]
    table = []
    for r in rows[1:]:
        if not r or (len(r) == 1 and r[0] == ""):
            continue
        cells = [(r[i] if i < len(r) else "") for i in keep]
        table.append((r[0], [None if c in _NA else c for c in cells]))
    return strains, table


def load_genomes(names, gff_paths, fasta_paths=None, log=None):
    """prep_data_n_fasta, input.py:67-138 (log: parse_gff's warnings, `{e}, skipping line "..." from {file}`, :326-329)"""
    data = {}
    for i, nm in enumerate(names):
        feats, _w = parse_gff(gff_paths[i])
        if log is not None:
            log.extend(_w)
        if fasta_paths and fasta_paths[i]:
            contigs = read_fasta(open(fasta_paths[i]).read())
        else:
            contigs = read_fasta(open(gff_paths[i]).read().split("##FASTA")[1])
        data[nm] = (contigs, feats)
    return data


def iter_gene_clusters(strains, table, genome_data, up, down, down_start_codon, gene_list=None, log=None,
                       raise_missing=False):
    missing = set(strains).difference(genome_data.keys())
    if missing:
        if log is not None:
            log.append(f"There are {len(missing)} strains present in the pangenome table but not in the GFF directory")
        if raise_missing:
            raise KeyError(f"Missing {len(missing)} from the GFF directory")
    sorted_strains = sorted(strains)
    sortstrain = {x: i for i, x in enumerate(sorted_strains)}
    for idx, cells in table:
        if gene_list is not None and idx not in gene_list:
            continue
        gene_sequences = {}
        clusterpresab = np.zeros(len(strains), dtype=np.int64)
        for strain, genes in zip(strains, cells):
            if genes is not None:
                clusterpresab[sortstrain[strain]] = 1
        for strain, genes in zip(strains, cells):
            if genes is None or strain not in genome_data:
                continue
            gene_sequences[strain] = []
            contigs, features = genome_data[strain]
            for gene in genes.split(";"):
                feat = features.get(gene)
                if feat is None:
                    if log is not None:
                        log.append(f"Could not find gene {gene} from {idx} in {strain}")
                    if raise_missing:
                        raise KeyError(f"Could not find gene {gene} from {idx} in {strain}")
                    continue
                ctg = contigs.get(feat.chromosome)
                if ctg is None:
                    if log is not None:
                        log.append(f"Could not find chromosome {feat.chromosome} in {strain}")
                    if raise_missing:
                        raise KeyError(f"Could not find chromosome {feat.chromosome} in {strain}")
                    continue
                offset = feat.start - 1 if (feat.strand > 0 and feat.start - 1 - up < 0) else up
                offset_d = feat.start - 1 if (feat.strand < 0 and feat.start - 1 - down < 0) else down
                if not down_start_codon:
                    if feat.strand > 0:
                        seq = ctg[feat.start - 1 - offset:feat.end + offset_d]
                        seq_start, seq_end = feat.start - offset, feat.end + offset_d
                    else:
                        seq = ctg[feat.start - 1 - offset_d:feat.end + offset][::-1].translate(_COMP)
                        seq_start, seq_end = feat.start - offset_d, feat.end + offset
                else:
                    if feat.strand > 0:
                        seq = ctg[feat.start - 1 - offset:feat.start + offset_d]
                        seq_start, seq_end = feat.start - offset, feat.start + offset_d
                    else:
                        seq = ctg[feat.end - 1 - offset_d:feat.end + offset][::-1].translate(_COMP)
                        seq_start, seq_end = feat.end - offset_d, feat.end + offset
                gene_sequences[strain].append(Seqinfo(seq, seq.translate(_COMP), feat.id, feat.chromosome,
                                                      seq_start, seq_end, feat.strand, offset))
        # input.py:373 `absent = strains.difference(present)`: pandas sorts the difference -- except when `present` is
        # empty (a row with no gene at all), where Index.difference returns the index as it is: table order
        absent = [s for s, g in zip(strains, cells) if g is None]
        for strain in (sorted(absent) if len(absent) < len(strains) else absent):
            gene_sequences[strain] = []
        yield gene_sequences, idx, clusterpresab
