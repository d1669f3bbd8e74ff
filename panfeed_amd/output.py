"""Output file surface of the hot path (names, headers, gzip mode), as the reference fixes it in
/root/reference/panfeed/input.py:235-259 and /root/reference/panfeed/panfeed.py:116-129."""
import gzip
import os

from .engine import KMERS_TSV_HEADER, KMERS_TO_HASHES_HEADER, hashes_to_patterns_header


def create_kmer_stroi(output, compress=False):
    """kmers.tsv[.gz] with its header written (input.py:235-247)."""
    if not compress:
        fh = open(os.path.join(output, "kmers.tsv"), "w")
    else:
        fh = gzip.open(os.path.join(output, "kmers.tsv.gz"), "wt", compresslevel=9)
    fh.write(KMERS_TSV_HEADER)
    fh.flush()
    return fh


def create_hash_files(output, compress=False):
    """(hashes_to_patterns, kmers_to_hashes) handles, no headers yet (input.py:249-259)."""
    if not compress:
        hash_pat = open(os.path.join(output, "hashes_to_patterns.tsv"), "w")
        kmer_hash = open(os.path.join(output, "kmers_to_hashes.tsv"), "w")
    else:
        hash_pat = gzip.open(os.path.join(output, "hashes_to_patterns.tsv.gz"), "wt", compresslevel=9)
        kmer_hash = gzip.open(os.path.join(output, "kmers_to_hashes.tsv.gz"), "wt", compresslevel=9)
    return hash_pat, kmer_hash


def write_headers(hash_pat, kmer_hash, genepres):
    """panfeed.py:116-129; `genepres` only needs `.columns` (the strain names)."""
    hash_pat.write(hashes_to_patterns_header(list(genepres.columns)))
    hash_pat.flush()
    kmer_hash.write(KMERS_TO_HASHES_HEADER)
    kmer_hash.flush()
