// pf_ingest.h -- internal interface between the reader (pf_input.cpp, no GPU code) and the device side of
// pf_pangenome_open_device (pf_api.hip): the reader puts a genome's file text where the sink says (pinned host memory),
// finds out -- without copying a base -- where every contig's letters lie and how its lines are wrapped, and hands the
// sink one piece per contig; the sink's kernel de-wraps, upper-cases and packs them to 2 bits per base in the genome store.
// Not part of the C ABI (include/panfeed_hip.h).
#pragma once
#include <cstddef>
#include <cstdint>

struct pf_pangenome;
struct pf_pangenome_opts_tag;

struct pf_ingest_piece {
    uint64_t text_off;       // byte offset, inside the block, of the contig's first letter
    uint64_t nbases;         // letters of the contig (line ends not counted)
    uint64_t dst_word;       // its first word in the genome store (claimed from the sink)
    uint32_t width;          // letters per line; 0 = the letters lie back to back
    uint32_t eol;            // bytes between two lines (1 "\n", 2 "\r\n")
};

struct pf_ingest_sink {
    void* self;
    // a block of at least `bytes` bytes for one file's text; *slot names it
    int (*acquire)(void* self, size_t bytes, char** host, uint32_t* slot);
    // `nwords` words of the genome store (2 * ceil(len / 64) + 4 per contig); UINT64_MAX when the store is full
    uint64_t (*claim_words)(void* self, uint64_t nwords);
    // the block's text is in place: pack these pieces (asynchronous); the block goes back to the sink
    int (*submit)(void* self, uint32_t slot, size_t text_bytes, const pf_ingest_piece* pieces, uint32_t n);
};

// pf_pangenome_open with the genomes' letters going to the sink instead of into host strings (contig text is kept only
// for contigs with a letter other than A/C/G/T and for target strains); the reader comes back in by-reference mode.
int pf_pangenome_open_sink(const void* opts, pf_ingest_sink* sink, pf_pangenome** out);
