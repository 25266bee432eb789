// hal2maf's text on the device (gfx950): the blocks of a batch, as the block state machine's walk logged them, become MAF text in
// HBM and come to the host as text.
//
// The reference prints a block entry by entry, a character per column (MafBlock::operator<<, maf/impl/halMafBlock.cpp:499-520,
// after appendColumn / updateEntry, :114-138, :370-395).  The host's rendering threads did the same from the walk's log — which
// base of which event went to which entry — and config 3's 1.8 GB of text cost them 8 CPU seconds: on a host whose container may
// use sixteen CPUs that is half a second whatever the threads, and the rows' bases lie anywhere in the packed DNA (a cache miss a
// row).  Here the log goes to the device as it is (a few megabytes a batch) and two kernels make the text:
//
//   k_maf_render_blocks / _rowlen / _sizes   whose slot every remembered entry of every block is; a lane an entry: was it given a
//                        base (only those are rows), its row's length in characters (its sequence's name and length are text per
//                        sequence, its start and length are counted in digits); a lane a block: the rows' places in the block (the
//                        reference's row first), the block's length;
//   (exclusive scan of the blocks' lengths: the blocks' places in the batch's text)
//   k_maf_render_rows    a lane a remembered entry: a row's lane writes "s <name> <start> <length> <strand> <sequence length>
//                        <bases>" — a run of bases from the packed DNA (reverse strand: leftwards, complemented: halCommon.h:45-75,
//                        187-190) where an event gave the entry a base, gaps where it did not.
//
// The text is the rendering threads' text byte for byte (tests: every hal2maf golden and oracle comparison goes through it on a
// GPU box; HGX_MAF_DEVICE_RENDER=0 keeps the threads).
#pragma once
#include "hgx_column_kernels.hpp"
#include "hgx_columns_engine.hpp"

namespace hgx {

struct MafRenderParams {
    const MafRenderBlock *blocks;
    const int32_t *entRank;
    const MafRenderEvent *events;
    const uint32_t *rowEnt;
    const MafRenderRow *rows;
    const MafRenderRank *ranks;
    const char *chars;
    const GenomeDesc *desc;
    uint32_t numBlocks, slots;
    int keepEmptyRefBlocks;
    uint32_t *blockLen;       // [numBlocks] characters of the block (0: not written)
    const uint32_t *blockOff; // [numBlocks + 1] after the scan
    uint32_t *rowOff;         // [slots] a row's place in its block, MAF_NO_ROW: the entry has no row
    uint32_t *slotBlock;      // [slots] the block of a slot
    unsigned long long *total; // characters of the batch
    unsigned int *error;      // 1: a block of 2^32 characters or more
    char *text;
};
static constexpr uint32_t MAF_NO_ROW = 0xFFFFFFFFu;

HGX_DEV __forceinline__ uint32_t maf_digits(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10) {
        v /= 10;
        ++n;
    }
    return n;
}
// the row of event e that entry j was given, or -1
HGX_DEV __forceinline__ int32_t maf_cell(const MafRenderParams &P, const MafRenderEvent &ev, uint32_t j) {
    const uint32_t *idx = P.rowEnt + ev.firstIdx;
    for (uint32_t r = 0; r < ev.nRows; ++r)
        if (idx[r] == j)
            return (int32_t)r;
    return -1;
}
// start (key >> 1 of the first base the entry was given), strand and length of entry j's row; false: it was given none
HGX_DEV __forceinline__ bool maf_row_fields(const MafRenderParams &P, const MafRenderBlock &B, uint32_t j, int64_t &start, int64_t &length, bool &rev) {
    const MafRenderEvent *ev = P.events + B.firstEvent;
    start = -1;
    length = 0;
    rev = false;
    for (uint32_t e = 0; e < B.numEvents; ++e) {
        const int32_t r = maf_cell(P, ev[e], j);
        if (r < 0)
            continue;
        if (start < 0) {
            const int64_t key = P.rows[ev[e].rowsOff + (uint32_t)r].key;
            start = key >> 1;
            rev = (key & 1) != 0;
        }
        length += ev[e].k;
    }
    return start >= 0;
}

// ---- sizes: three small launches, the middle one a lane per remembered ENTRY (a lane per block going through the block's 29
// entries one after the other was a millisecond a batch: 34 k lanes on a device that holds half a million) ----
// a lane a block: whose slots these are, the block's columns
static __global__ void __launch_bounds__(256) k_maf_render_blocks(MafRenderParams P) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < P.numBlocks; b += gridDim.x * blockDim.x) {
        const MafRenderBlock B = P.blocks[b];
        for (uint32_t j = 0; j < B.numEnts; ++j)
            P.slotBlock[B.slotBase + j] = b;
    }
}
// a lane an entry of a block: the length of its row (rowOff[s], for the moment), 0: it has none
HGX_DEV __forceinline__ void maf_render_rowlen_body(const MafRenderParams &P, uint32_t s) {
    const uint32_t b = P.slotBlock[s];
    const MafRenderBlock B = P.blocks[b];
    const uint32_t j = s - B.slotBase;
    const MafRenderEvent *ev = P.events + B.firstEvent;
    int64_t start, length;
    bool rev;
    bool given = maf_row_fields(P, B, j, start, length, rev);
    if (!given && B.refEnt >= 0 && j == (uint32_t)B.refEnt && B.refIndex != -1 && P.keepEmptyRefBlocks) { // (a row of gaps, as long as the block)
        given = true;
        start = B.refIndex;
        length = 0;
    }
    uint32_t len = 0;
    if (given) {
        uint64_t columns = 0;
        for (uint32_t e = 0; e < B.numEvents; ++e)
            columns += (uint64_t)ev[e].k;
        const MafRenderRank R = P.ranks[P.entRank[B.firstEnt + j]];
        const uint64_t n = (uint64_t)R.headLen + maf_digits((uint64_t)start) + 1 + maf_digits((uint64_t)length) + 2 + R.tailLen + columns + 1;
        if (n >= 0x7FFFFFFFull)
            *P.error = 1;
        else
            len = (uint32_t)n;
    }
    P.rowOff[s] = len;
}
static __global__ void __launch_bounds__(256) k_maf_render_rowlen(MafRenderParams P) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < P.slots; s += gridDim.x * blockDim.x)
        maf_render_rowlen_body(P, s);
}
// a lane a block: the rows' places in the block — MafBlock's operator<< (halMafBlock.cpp:499-520): the reference's row first, then
// the entries that have a start — and the block's length; a block whose reference was given no base is not written unless empty
// reference blocks are kept (referenceIsAllGaps, halMafExport.cpp:70, 85)
HGX_DEV __forceinline__ void maf_render_sizes_body(const MafRenderParams &P, uint32_t b) {
    const MafRenderBlock B = P.blocks[b];
    uint64_t off = 0;
    bool written = false;
    if (B.refEnt >= 0) {
        const uint32_t ref = (uint32_t)B.refEnt;
        int64_t start, length;
        bool rev;
        written = P.keepEmptyRefBlocks || maf_row_fields(P, B, ref, start, length, rev);
        if (written) {
            off = 2; // "a\n"
            const uint32_t refLen = P.rowOff[B.slotBase + ref];
            if (refLen) {
                P.rowOff[B.slotBase + ref] = (uint32_t)off;
                off += refLen;
            } else {
                P.rowOff[B.slotBase + ref] = MAF_NO_ROW;
            }
            for (uint32_t j = 0; j < B.numEnts; ++j) {
                if (j == ref)
                    continue;
                const uint32_t len = P.rowOff[B.slotBase + j];
                if (len && off < 0xFFFFFFF0ull) {
                    P.rowOff[B.slotBase + j] = (uint32_t)off;
                    off += len;
                } else {
                    P.rowOff[B.slotBase + j] = MAF_NO_ROW;
                    off += len;
                }
            }
            off += 1; // the empty line behind the block
        }
    }
    if (!written)
        for (uint32_t j = 0; j < B.numEnts; ++j)
            P.rowOff[B.slotBase + j] = MAF_NO_ROW;
    if (off >= 0xFFFFFFF0ull) {
        *P.error = 1;
        off = 0;
    }
    P.blockLen[b] = (uint32_t)off;
    if (off)
        atomicAdd(P.total, (unsigned long long)off);
}
static __global__ void __launch_bounds__(256) k_maf_render_sizes(MafRenderParams P) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < P.numBlocks; b += gridDim.x * blockDim.x)
        maf_render_sizes_body(P, b);
}

HGX_DEV __forceinline__ char *maf_put_number(char *o, uint64_t v) {
    char tmp[20];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + (int)(v % 10));
        v /= 10;
    } while (v);
    while (n > 0)
        *o++ = tmp[--n];
    return o;
}
HGX_DEV __forceinline__ void maf_render_row_body(const MafRenderParams &P, uint32_t s) {
    const uint32_t b = P.slotBlock[s];
    const MafRenderBlock B = P.blocks[b];
    const uint32_t j = s - B.slotBase;
    const uint32_t len = P.blockLen[b];
    char *const blockText = P.text + P.blockOff[b];
    if (j == 0 && len) {
        blockText[0] = 'a';
        blockText[1] = '\n';
        blockText[len - 1] = '\n';
    }
    const uint32_t at = P.rowOff[s];
    if (at == MAF_NO_ROW || !len)
        return;
    int64_t start, length;
    bool rev;
    const bool given = maf_row_fields(P, B, j, start, length, rev);
    if (!given) { // (the reference's row of a block that gave it no base)
        start = B.refIndex;
        length = 0;
        rev = false;
    }
    const MafRenderRank R = P.ranks[P.entRank[B.firstEnt + j]];
    char *o = blockText + at;
    for (uint32_t i = 0; i < R.headLen; ++i)
        *o++ = P.chars[R.headOff + i];
    o = maf_put_number(o, (uint64_t)start);
    *o++ = '\t';
    o = maf_put_number(o, (uint64_t)length);
    *o++ = '\t';
    *o++ = rev ? '-' : '+';
    for (uint32_t i = 0; i < R.tailLen; ++i)
        *o++ = P.chars[R.tailOff + i];
    const uint8_t *dna = P.desc[R.genome].dna;
    const MafRenderEvent *ev = P.events + B.firstEvent;
    for (uint32_t e = 0; e < B.numEvents; ++e) {
        const int64_t k = ev[e].k;
        const int32_t r = given ? maf_cell(P, ev[e], j) : -1;
        if (r < 0) {
            for (int64_t i = 0; i < k; ++i)
                *o++ = '-';
            continue;
        }
        const int64_t key = P.rows[ev[e].rowsOff + (uint32_t)r].key;
        const bool rv = (key & 1) != 0;
        // (the base's genome coordinate: the packed DNA is read there)
        const int64_t pos = R.seqStart + (rv ? R.srcLength - 1 - (key >> 1) : key >> 1);
        if (!dna) {
            for (int64_t i = 0; i < k; ++i)
                *o++ = 'N';
        } else if (!rv) { // dnaUnpack (halCommon.h:187-190): two bases a byte, the even one in the high half
            for (int64_t i = 0; i < k; ++i) {
                const int64_t p = pos + i;
                const uint8_t byte = dna[p >> 1];
                const int nib = (p & 1) ? (byte & 0x0F) : (byte >> 4);
                *o++ = "acgtn\0\0\0ACGTN\0\0\0"[nib];
            }
        } else { // the reverse strand: leftwards, complemented (reverseComplement, halCommon.h:45-75)
            for (int64_t i = 0; i < k; ++i) {
                const int64_t p = pos - i;
                const uint8_t byte = dna[p >> 1];
                const int nib = (p & 1) ? (byte & 0x0F) : (byte >> 4);
                *o++ = "tgcan\0\0\0TGCAN\0\0\0"[nib];
            }
        }
    }
    *o++ = '\n';
}
static __global__ void __launch_bounds__(256) k_maf_render_rows(MafRenderParams P) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < P.slots; s += gridDim.x * blockDim.x)
        maf_render_row_body(P, s);
}

} // namespace hgx
