// extern "C" boundary of libhgx (include/hgx.h).  No exception crosses it.
#include "../../include/hgx.h"
#include "hgx_textmem.hpp"
#include "hgx_columns_host.hpp"
#include "hgx_liftover_host.hpp"
#include <cstdlib>
#include <cstring>
#include <sstream>

using namespace hgx;

static void setErr(char **err, const std::string &msg) {
    if (err) {
        *err = (char *)malloc(msg.size() + 1);
        if (*err)
            memcpy(*err, msg.c_str(), msg.size() + 1);
    }
}

#define HGX_TRY try {
#define HGX_CATCH                                                                                                      \
    }                                                                                                                  \
    catch (std::exception & e) {                                                                                       \
        setErr(err, e.what());                                                                                         \
        return HGX_ERR;                                                                                                \
    }                                                                                                                  \
    catch (...) {                                                                                                      \
        setErr(err, "unknown error");                                                                                  \
        return HGX_ERR;                                                                                                \
    }

struct hgx_builder {
    Image img;
};

static int finishHandle(Image &&img, int device, hgx_alignment **out) {
    std::unique_ptr<hgx_alignment> h(new hgx_alignment);
    h->img = std::move(img);
    if (h->img.newick.empty())
        h->img.newick = h->img.buildNewick();
    h->img.validate();
    if (device >= 0)
        h->dev = uploadImage(h->img, device);
    *out = h.release();
    return HGX_OK;
}

extern "C" {

int hgx_open(const char *path, int device, hgx_alignment **out, char **err) {
    HGX_TRY
    if (!path || !out)
        throw std::runtime_error("hgx_open: null argument");
    return finishHandle(openAlignmentFile(path), device, out);
    HGX_CATCH
}

void hgx_close(hgx_alignment *h) {
    delete h;
}

int hgx_clone_to_device(const hgx_alignment *h, int device, hgx_alignment **out, char **err) {
    HGX_TRY
    if (!h || !out)
        throw std::runtime_error("hgx_clone_to_device: null argument");
    if (device < 0)
        throw std::runtime_error("hgx_clone_to_device: a device ordinal is needed");
    std::unique_ptr<hgx_alignment> c(new hgx_alignment(h->imgHolder));
    c->dev = uploadImage(c->img, device);
    *out = c.release();
    return HGX_OK;
    HGX_CATCH
}

int hgx_builder_begin(hgx_builder **out, char **err) {
    HGX_TRY
    *out = new hgx_builder;
    return HGX_OK;
    HGX_CATCH
}

int hgx_builder_add_genome(hgx_builder *b, const char *name, const char *parent_name, double branch_length, int64_t num_sequences,
                           const char *const *seq_names, const int64_t *seq_lengths, const int64_t *seq_num_top,
                           const int64_t *seq_num_bottom, int64_t num_top, const int64_t *top_start, const int64_t *top_parent_index,
                           const uint8_t *top_parent_reversed, const int64_t *top_next_paralogy, const int64_t *top_bottom_parse,
                           int64_t num_bottom, const int64_t *bottom_start, const int64_t *bottom_top_parse, int64_t num_children,
                           const int64_t *child_index, const uint8_t *child_reversed, const char *dna, char **err) {
    HGX_TRY
    if (!b || !name)
        throw std::runtime_error("hgx_builder_add_genome: null argument");
    Image &img = b->img;
    if (img.genomeByName(name) >= 0)
        throw std::runtime_error(std::string("genome added twice: ") + name);
    GenomeTables G;
    G.name = name;
    G.branchLength = branch_length;
    if (parent_name) {
        G.parent = img.genomeByName(parent_name);
        if (G.parent < 0)
            throw std::runtime_error(std::string("parent genome must be added before its child: ") + parent_name);
    } else if (!img.genomes.empty()) {
        throw std::runtime_error("only the first genome added may be the root");
    }
    if (num_sequences < 0 || num_top < 0 || num_bottom < 0 || num_children < 0)
        throw std::runtime_error("hgx_builder_add_genome: negative count");
    if ((num_sequences > 0 && (!seq_names || !seq_lengths)) ||
        (num_top > 0 && (!top_start || !top_parent_index || !top_parent_reversed || !top_next_paralogy)) || (num_bottom > 0 && !bottom_start) ||
        (num_children > 0 && num_bottom > 0 && (!child_index || !child_reversed)))
        throw std::runtime_error("hgx_builder_add_genome: null array");
    int64_t pos = 0, ti = 0, bi = 0;
    for (int64_t s = 0; s < num_sequences; ++s) {
        if (!seq_names[s] || seq_lengths[s] < 0 || (seq_num_top && seq_num_top[s] < 0) || (seq_num_bottom && seq_num_bottom[s] < 0))
            throw std::runtime_error("hgx_builder_add_genome: bad sequence entry");
        SeqInfo S;
        S.name = seq_names[s];
        S.start = pos;
        S.length = seq_lengths[s];
        S.topStart = ti;
        S.numTop = seq_num_top ? seq_num_top[s] : 0;
        S.botStart = bi;
        S.numBot = seq_num_bottom ? seq_num_bottom[s] : 0;
        pos += S.length;
        ti += S.numTop;
        bi += S.numBot;
        G.seqs.push_back(S);
    }
    if ((seq_num_top && ti != num_top) || (seq_num_bottom && bi != num_bottom))
        throw std::runtime_error("hgx_builder_add_genome: the sequences' segment counts do not add up to the genome's");
    G.totalLength = pos;
    G.numTop = num_top;
    G.numBot = num_bottom;
    G.tStart.assign(top_start, top_start + (num_top > 0 ? num_top + 1 : 0));
    if (num_top == 0)
        G.tStart.assign(1, pos);
    G.tParent.assign(top_parent_index, top_parent_index + num_top);
    G.tParentRev.assign(top_parent_reversed, top_parent_reversed + num_top);
    G.tParalogy.assign(top_next_paralogy, top_next_paralogy + num_top);
    if (top_bottom_parse)
        G.tBotParse.assign(top_bottom_parse, top_bottom_parse + num_top);
    else
        G.tBotParse.assign((size_t)num_top, NULL_INDEX);
    G.bStart.assign(bottom_start, bottom_start + (num_bottom > 0 ? num_bottom + 1 : 0));
    if (num_bottom == 0)
        G.bStart.assign(1, pos);
    if (bottom_top_parse)
        G.bTopParse.assign(bottom_top_parse, bottom_top_parse + num_bottom);
    else
        G.bTopParse.assign((size_t)num_bottom, NULL_INDEX);
    G.bChild.resize((size_t)num_children);
    G.bChildRev.resize((size_t)num_children);
    for (int64_t k = 0; k < num_children; ++k) {
        G.bChild[(size_t)k].assign(child_index + k * num_bottom, child_index + (k + 1) * num_bottom);
        G.bChildRev[(size_t)k].assign(child_reversed + k * num_bottom, child_reversed + (k + 1) * num_bottom);
    }
    if (dna)
        packDna(std::string(dna, (size_t)pos), G.dna);
    const int id = (int)img.genomes.size();
    if (G.parent >= 0)
        img.genomes[(size_t)G.parent].children.push_back(id);
    img.genomes.push_back(std::move(G));
    return HGX_OK;
    HGX_CATCH
}

int hgx_builder_finish(hgx_builder *b, int device, hgx_alignment **out, char **err) {
    std::unique_ptr<hgx_builder> guard(b);
    HGX_TRY
    if (!b || !out)
        throw std::runtime_error("hgx_builder_finish: null argument");
    for (GenomeTables &G : b->img.genomes)
        if (G.children.size() != G.bChild.size())
            throw std::runtime_error("genome " + G.name + ": " + std::to_string(G.bChild.size()) + " child slots declared but " +
                                     std::to_string(G.children.size()) + " children added");
    return finishHandle(std::move(b->img), device, out);
    HGX_CATCH
}

void hgx_builder_abort(hgx_builder *b) {
    delete b;
}

int hgx_num_genomes(const hgx_alignment *h) {
    return (int)h->img.genomes.size();
}
const char *hgx_newick(const hgx_alignment *h) {
    return h->img.newick.c_str();
}
static const GenomeTables *genomeOf(const hgx_alignment *h, int g) {
    if (!h || g < 0 || g >= (int)h->img.genomes.size())
        return nullptr;
    return &h->img.genomes[(size_t)g];
}
const char *hgx_genome_name(const hgx_alignment *h, int g) {
    const GenomeTables *G = genomeOf(h, g);
    return G ? G->name.c_str() : nullptr;
}
int hgx_genome_id(const hgx_alignment *h, const char *name) {
    return (h && name) ? h->img.genomeByName(name) : -1;
}
int hgx_genome_parent(const hgx_alignment *h, int g) {
    const GenomeTables *G = genomeOf(h, g);
    return G ? G->parent : -1;
}
int hgx_genome_num_children(const hgx_alignment *h, int g) {
    const GenomeTables *G = genomeOf(h, g);
    return G ? (int)G->children.size() : -1;
}
int hgx_genome_child(const hgx_alignment *h, int g, int k) {
    const GenomeTables *G = genomeOf(h, g);
    return (G && k >= 0 && k < (int)G->children.size()) ? G->children[(size_t)k] : -1;
}
int64_t hgx_genome_length(const hgx_alignment *h, int g) {
    const GenomeTables *G = genomeOf(h, g);
    return G ? G->totalLength : -1;
}
int64_t hgx_genome_num_top(const hgx_alignment *h, int g) {
    const GenomeTables *G = genomeOf(h, g);
    return G ? G->numTop : -1;
}
int64_t hgx_genome_num_bottom(const hgx_alignment *h, int g) {
    const GenomeTables *G = genomeOf(h, g);
    return G ? G->numBot : -1;
}
int hgx_genome_num_sequences(const hgx_alignment *h, int g) {
    const GenomeTables *G = genomeOf(h, g);
    return G ? (int)G->seqs.size() : -1;
}
int hgx_sequence_info(const hgx_alignment *h, int g, int s, const char **name, int64_t *start, int64_t *length) {
    const GenomeTables *G = genomeOf(h, g);
    if (!G || s < 0 || s >= (int)G->seqs.size())
        return HGX_ERR;
    if (name)
        *name = G->seqs[(size_t)s].name.c_str();
    if (start)
        *start = G->seqs[(size_t)s].start;
    if (length)
        *length = G->seqs[(size_t)s].length;
    return HGX_OK;
}
int hgx_sequence_lookup(const hgx_alignment *h, int g, const char *name, int64_t *start, int64_t *length) {
    const GenomeTables *G = genomeOf(h, g);
    if (!G || !name)
        return -1;
    const int s = G->seqIndexByName(name);
    if (s >= 0) {
        if (start)
            *start = G->seqs[(size_t)s].start;
        if (length)
            *length = G->seqs[(size_t)s].length;
    }
    return s;
}
int hgx_mrca(const hgx_alignment *h, int a, int b) {
    if (!genomeOf(h, a) || !genomeOf(h, b))
        return -1;
    return h->img.lca(a, b);
}

static hgx_liftover_opts defaultOpts(const hgx_liftover_opts *o) {
    hgx_liftover_opts d;
    d.traverse_dupes = 1;
    d.coalescence_limit = -1;
    d.min_length = 0;
    d.emit_blocks = 0;
    d.block_mapper_source = 0;
    return o ? *o : d;
}

int hgx_liftover_batch(hgx_alignment *h, int src, int tgt, size_t n, const hgx_interval *iv, const hgx_liftover_opts *opts,
                       hgx_record **out, size_t *n_out, char **err) {
    HGX_TRY
    if (!h || !out || !n_out || (n && !iv))
        throw std::runtime_error("hgx_liftover_batch: null argument");
    if (!genomeOf(h, src) || !genomeOf(h, tgt))
        throw std::runtime_error("hgx_liftover_batch: genome id out of range");
    *out = nullptr;
    *n_out = 0;
    try {
        liftoverBatchHostRaw(h, src, tgt, n, iv, defaultOpts(opts), [&](size_t nrec) {
            *out = (hgx_record *)malloc(std::max<size_t>(1, nrec) * sizeof(hgx_record));
            if (!*out)
                throw std::runtime_error("out of memory");
            *n_out = nrec;
            return *out;
        });
    } catch (...) { // (the copy from the device can still fail after the buffer was handed out)
        free(*out);
        *out = nullptr;
        *n_out = 0;
        throw;
    }
    return HGX_OK;
    HGX_CATCH
}

int hgx_block_map(hgx_alignment *h, int ref, int query, int64_t abs_ref_first, int64_t abs_ref_last, int target_reversed, int do_dupes,
                  int64_t min_length, int coalescence_limit, hgx_record **out, size_t *n_out, char **err) {
    HGX_TRY
    if (!h || !out || !n_out)
        throw std::runtime_error("hgx_block_map: null argument");
    if (!genomeOf(h, ref) || !genomeOf(h, query))
        throw std::runtime_error("hgx_block_map: genome id out of range");
    hgx_liftover_opts o = defaultOpts(nullptr);
    o.traverse_dupes = do_dupes ? 1 : 0;
    o.min_length = min_length;
    o.coalescence_limit = coalescence_limit;
    std::vector<hgx_record> recs;
    blockMapHost(h, ref, query, abs_ref_first, abs_ref_last, target_reversed != 0, o, recs);
    *out = (hgx_record *)malloc(std::max<size_t>(1, recs.size()) * sizeof(hgx_record));
    if (!*out)
        throw std::runtime_error("out of memory");
    if (!recs.empty())
        memcpy(*out, recs.data(), recs.size() * sizeof(hgx_record));
    *n_out = recs.size();
    return HGX_OK;
    HGX_CATCH
}

void hgx_free_block_results(hgx_block_results *results) { // halFreeBlockResults (blockViz/impl/halBlockViz.cpp:207-241)
    if (!results)
        return;
    for (hgx_block *b = results->mappedBlocks; b;) {
        hgx_block *next = b->next;
        free(b->qChrom);
        free(b->qSequence);
        free(b->tSequence);
        free(b);
        b = next;
    }
    for (hgx_target_dupe_list *d = results->targetDupeBlocks; d;) {
        hgx_target_dupe_list *next = d->next;
        for (hgx_target_range *r = d->tRange; r;) {
            hgx_target_range *rn = r->next;
            free(r);
            r = rn;
        }
        free(d->qChrom);
        free(d);
        d = next;
    }
    free(results);
}

namespace {
struct ArgumentMessage : std::runtime_error { // halGetBlocksInTargetRange's own argument checks: reported as they are (halBlockViz.cpp:251-265)
    using std::runtime_error::runtime_error;
};
} // namespace

int hgx_get_blocks_in_target_ranges(hgx_alignment *h, const char *q_species, const char *t_species, const char *t_chrom, size_t n,
                                    const int64_t *t_starts, const int64_t *t_ends, int64_t t_reversed, int seq_mode, int dup_mode,
                                    int map_back_adjacencies, const char *coalescence_limit_name, hgx_block_results **results, char **err) {
    std::vector<hgx_block_results *> out;
    if (results)
        for (size_t k = 0; k < n; ++k)
            results[k] = nullptr;
    try {
        if (!h || !q_species || !t_species || !t_chrom || !results || (n && (!t_starts || !t_ends)))
            throw std::runtime_error("hgx_get_blocks_in_target_ranges: null argument");
        if (!h->dev)
            throw std::runtime_error("alignment was opened without a device (device = -1); the block mapper needs the HIP path");
        // halGetBlocksInTargetRange's own checks, with its messages (halBlockViz.cpp:251-302; checkGenomes :716-738)
        for (size_t k = 0; k < n; ++k) // (the range check comes first in the reference, :251-257)
            if (t_ends[k] - t_starts[k] < 0)
                throw ArgumentMessage("halGetBlocksInTargetRange invalid query range [" + std::to_string(t_starts[k]) + "," + std::to_string(t_ends[k]) + ")");
        if (t_reversed != 0 && map_back_adjacencies != 0)
            throw ArgumentMessage("halGetBlocksInTargetRange tReversed can only be set when mapBackAdjacencies is 0");
        if (t_reversed != 0 && dup_mode == 2)
            throw ArgumentMessage("tReversed cannot be set in conjunction with dupMode=HAL_QUERY_AND_TARGET_DUPS");
        const Image &img = h->img;
        const int q = img.genomeByName(q_species), t = img.genomeByName(t_species);
        if (q < 0)
            throw std::runtime_error(std::string("Query species ") + q_species + " not found in alignment");
        if (t < 0)
            throw std::runtime_error(std::string("Reference species ") + t_species + " not found in alignment");
        const GenomeTables &T = img.genomes[(size_t)t];
        const int s = T.seqIndexByName(t_chrom);
        if (s < 0)
            throw std::runtime_error(std::string("Unable to locate sequence ") + t_chrom + " in genome " + t_species);
        int limit = -1;
        if (coalescence_limit_name) {
            limit = img.genomeByName(coalescence_limit_name);
            if (limit < 0)
                throw std::runtime_error(std::string("Could not find coalescence limit ") + coalescence_limit_name + " in alignment");
            // (the reference finds out while it climbs, halSegmentMapper.cpp:541; an ancestor of the MRCA is what it can use)
            int g = img.lca(q, t);
            while (g >= 0 && g != limit)
                g = img.genomes[(size_t)g].parent;
            if (g < 0)
                throw std::runtime_error("Hit root genome when attempting to map paralogies");
        }
        const SeqInfo &S = T.seqs[(size_t)s];
        std::vector<std::pair<int64_t, int64_t>> ranges;
        for (size_t k = 0; k < n; ++k) {
            const int64_t tStart = t_starts[k], tEnd = t_ends[k];
            const int64_t myEnd = tEnd > 0 ? tEnd : S.length;
            const int64_t absStart = S.start + tStart, absEnd = S.start + myEnd - 1;
            if (absStart > absEnd || tStart < 0)
                throw ArgumentMessage("halGetBlocksInTargetRange invalid range");
            if (absEnd > S.start + S.length - 1)
                throw ArgumentMessage("halGetBlocksInTargetRange target end position outside of target sequence");
            ranges.emplace_back(absStart, absEnd);
        }
        hgx::blocksInTargetRanges(h, q, t, ranges, t_reversed != 0, seq_mode != 0, dup_mode != 0, dup_mode == 2, map_back_adjacencies != 0, limit, out);
        for (size_t k = 0; k < n; ++k)
            results[k] = out[k];
        return HGX_OK;
    } catch (ArgumentMessage &e) {
        setErr(err, e.what());
        return HGX_ERR;
    } catch (std::exception &e) {
        for (hgx_block_results *r : out)
            hgx_free_block_results(r);
        setErr(err, std::string("halGetBlocksInTargetRange error reading blocks: ") + e.what());
        return HGX_ERR;
    } catch (...) {
        for (hgx_block_results *r : out)
            hgx_free_block_results(r);
        setErr(err, "halGetBlocksInTargetRange error reading blocks: unknown exception");
        return HGX_ERR;
    }
}

hgx_block_results *hgx_get_blocks_in_target_range(hgx_alignment *h, const char *q_species, const char *t_species, const char *t_chrom,
                                                  int64_t t_start, int64_t t_end, int64_t t_reversed, int seq_mode, int dup_mode,
                                                  int map_back_adjacencies, const char *coalescence_limit_name, char **err) {
    hgx_block_results *r = nullptr;
    if (hgx_get_blocks_in_target_ranges(h, q_species, t_species, t_chrom, 1, &t_start, &t_end, t_reversed, seq_mode, dup_mode, map_back_adjacencies,
                                        coalescence_limit_name, &r, err) != HGX_OK)
        return nullptr;
    return r;
}

int hgx_liftover_plan_create(hgx_alignment *h, int src, int tgt, const hgx_liftover_opts *opts, size_t max_queries,
                             hgx_liftover_plan **out, char **err) {
    HGX_TRY
    if (!h || !out)
        throw std::runtime_error("hgx_liftover_plan_create: null argument");
    *out = createLiftoverPlan(h, src, tgt, defaultOpts(opts), max_queries);
    return HGX_OK;
    HGX_CATCH
}

void hgx_liftover_plan_destroy(hgx_liftover_plan *p) {
    try {
        destroyLiftoverPlan(p);
    } catch (...) {
    }
}

int hgx_liftover_run_device(hgx_liftover_plan *p, size_t n, const int64_t *d_gstart, const int64_t *d_gend, const uint8_t *d_strand,
                            void *hip_stream, const hgx_record **d_records, size_t *n_records, char **err) {
    HGX_TRY
    if (!p || !d_records || !n_records)
        throw std::runtime_error("hgx_liftover_run_device: null argument");
    runLiftoverPlan(p, n, d_gstart, d_gend, d_strand, hip_stream, d_records, n_records);
    return HGX_OK;
    HGX_CATCH
}

int hgx_liftover_submit(hgx_liftover_plan *p, size_t n, const int64_t *d_gstart, const int64_t *d_gend, const uint8_t *d_strand, void *hip_stream,
                        char **err) {
    HGX_TRY
    if (!p)
        throw std::runtime_error("hgx_liftover_submit: null argument");
    submitLiftoverPlan(p, n, d_gstart, d_gend, d_strand, hip_stream);
    return HGX_OK;
    HGX_CATCH
}

int hgx_liftover_collect(hgx_liftover_plan *p, const hgx_record **d_records, size_t *n_records, char **err) {
    HGX_TRY
    if (!p || !d_records || !n_records)
        throw std::runtime_error("hgx_liftover_collect: null argument");
    collectLiftoverPlan(p, d_records, n_records);
    return HGX_OK;
    HGX_CATCH
}

int hgx_liftover_last_stats(const hgx_liftover_plan *p, hgx_liftover_stats *out) {
    if (!p || !out)
        return HGX_ERR;
    *out = liftoverPlanStats(p);
    return HGX_OK;
}

int hgx_liftover_copy_records(const hgx_liftover_plan *p, void *d_dst, size_t n_records, void *hip_stream, char **err) {
    HGX_TRY
    if (!p || (n_records && !d_dst))
        throw std::runtime_error("hgx_liftover_copy_records: null argument");
    liftoverPlanCopyRecords(p, d_dst, n_records, hip_stream);
    return HGX_OK;
    HGX_CATCH
}

int hgx_liftover_copy_records_packed(const hgx_liftover_plan *p, void *d_dst, size_t n_records, void *hip_stream, char **err) {
    HGX_TRY
    if (!p || (n_records && !d_dst))
        throw std::runtime_error("hgx_liftover_copy_records_packed: null argument");
    liftoverPlanCopyRecordsPacked(p, d_dst, n_records, hip_stream);
    return HGX_OK;
    HGX_CATCH
}

int hgx_liftover_wire_blob(hgx_liftover_plan *p, void *d_dst, size_t capacity, int64_t first_query, size_t *bytes, int *format, void *hip_stream,
                           char **err) {
    HGX_TRY
    if (!p || !bytes)
        throw std::runtime_error("hgx_liftover_wire_blob: null argument");
    *bytes = liftoverPlanWireBlob(p, d_dst, capacity, first_query, format, hip_stream);
    return HGX_OK;
    HGX_CATCH
}

int hgx_liftover_kernel_times(hgx_liftover_plan *p, char **json) {
    if (!p || !json)
        return HGX_ERR;
    try {
        const std::string s = liftoverPlanKernelTimes(p);
        *json = (char *)malloc(s.size() + 1);
        if (!*json)
            return HGX_ERR;
        memcpy(*json, s.c_str(), s.size() + 1);
        return HGX_OK;
    } catch (std::exception &e) { // (no error slot in this call's signature: the reason goes to stderr)
        fprintf(stderr, "hgx_liftover_kernel_times: %s\n", e.what());
        return HGX_ERR;
    } catch (...) {
        return HGX_ERR;
    }
}

int hgx_liftover_build_phases(char **json) {
    if (!json)
        return HGX_ERR;
    try {
        const std::string s = liftoverBuildPhases();
        *json = (char *)malloc(s.size() + 1);
        if (!*json)
            return HGX_ERR;
        memcpy(*json, s.c_str(), s.size() + 1);
        return HGX_OK;
    } catch (...) {
        return HGX_ERR;
    }
}

int hgx_liftover_plan_set_timing(hgx_liftover_plan *p, int mode) {
    if (!p)
        return HGX_ERR;
    try {
        liftoverPlanSetTiming(p, mode);
        return HGX_OK;
    } catch (std::exception &e) { // (no error slot in this call's signature: the reason goes to stderr)
        fprintf(stderr, "hgx_liftover_plan_set_timing: %s\n", e.what());
        return HGX_ERR;
    } catch (...) {
        return HGX_ERR;
    }
}

int hgx_liftover_plan_set_workers(hgx_liftover_plan *p, int n) {
    if (!p)
        return HGX_ERR;
    liftoverPlanSetWorkers(p, n);
    return HGX_OK;
}

static int convertOver(hgx_alignment *const *handles, int n_handles, int src, const char *bed_text, size_t bed_len, int tgt, int bed_type,
                       int traverse_dupes, int out_psl, int out_psl_with_name, int coalescence_limit, char **out_text, size_t *out_len, char **err);

int hgx_liftover_convert(hgx_alignment *h, int src, const char *bed_text, size_t bed_len, int tgt, int bed_type, int traverse_dupes,
                         int out_psl, int out_psl_with_name, int coalescence_limit, char **out_text, size_t *out_len, char **err) {
    return convertOver(&h, 1, src, bed_text, bed_len, tgt, bed_type, traverse_dupes, out_psl, out_psl_with_name, coalescence_limit, out_text, out_len,
                       err);
}

int hgx_liftover_render_blobs(hgx_alignment *h, int src, int tgt, const char *src_bed, size_t src_len, int bed_type, const void *const *blobs,
                              const size_t *blob_bytes, int n_blobs, char **out_text, size_t *out_len, char **err) {
    try {
        if (!h || !out_text || !out_len || (src_len && !src_bed) || n_blobs < 0 || (n_blobs && (!blobs || !blob_bytes)))
            throw std::runtime_error("hgx_liftover_render_blobs: null argument");
        if (!genomeOf(h, src) || !genomeOf(h, tgt))
            throw std::runtime_error("hgx_liftover_render_blobs: genome id out of range");
        liftoverRenderBlobs(h, src, tgt, src_bed ? src_bed : "", src_len, bed_type, blobs, blob_bytes, n_blobs, out_text, out_len);
        return HGX_OK;
    } catch (std::exception &e) {
        setErr(err, e.what());
        return HGX_ERR;
    }
}

int hgx_liftover_convert_multi(hgx_alignment *const *handles, int n_handles, int src, const char *bed_text, size_t bed_len, int tgt, int bed_type,
                               int traverse_dupes, int out_psl, int out_psl_with_name, int coalescence_limit, char **out_text, size_t *out_len,
                               char **err) {
    if (!handles || n_handles < 1) {
        setErr(err, "hgx_liftover_convert_multi: no handles");
        return HGX_ERR;
    }
    return convertOver(handles, n_handles, src, bed_text, bed_len, tgt, bed_type, traverse_dupes, out_psl, out_psl_with_name, coalescence_limit,
                       out_text, out_len, err);
}

static int convertOver(hgx_alignment *const *handles, int n_handles, int src, const char *bed_text, size_t bed_len, int tgt, int bed_type,
                       int traverse_dupes, int out_psl, int out_psl_with_name, int coalescence_limit, char **out_text, size_t *out_len, char **err) {
    hgx_alignment *h = handles[0];
    // Output produced before a failing input line is still returned (the reference has already written it
    // to the stream when it throws, halBedScanner.cpp:49-58).
    char *text = nullptr;
    size_t n = 0;
    int rc = HGX_OK;
    try {
        if (!h || !out_text || !out_len || (bed_len && !bed_text))
            throw std::runtime_error("hgx_liftover_convert: null argument");
        if (!genomeOf(h, src) || !genomeOf(h, tgt))
            throw std::runtime_error("hgx_liftover_convert: genome id out of range");
        Liftover lo;
        for (int i = 1; i < n_handles; ++i) {
            bool clone = handles[i] && handles[i]->imgHolder == h->imgHolder && handles[i]->dev;
#ifdef HGX_HOST_PROFILE
            // (the profiling build's replay, hgx_lift_replay.hpp: the one device-less handle several times over stands for the clones)
            if (getenv("HGX_LIFT_REPLAY"))
                clone = handles[i] && handles[i]->imgHolder == h->imgHolder;
#endif
            if (!clone)
                throw std::runtime_error("hgx_liftover_convert_multi: every handle must be a device clone of the first (hgx_clone_to_device)");
            lo.moreDevices.push_back(handles[i]);
        }
        lo.convertBuffer(h, src, bed_text ? bed_text : "", bed_len, tgt, &text, &n, bed_type, traverse_dupes != 0, out_psl != 0, out_psl_with_name != 0,
                         coalescence_limit);
    } catch (std::exception &e) {
        setErr(err, e.what());
        rc = HGX_ERR;
    } catch (...) {
        setErr(err, "unknown error");
        rc = HGX_ERR;
    }
    if (out_text && out_len) {
        if (!text) { // (nothing was lifted: an empty text, still released with hgx_free)
            text = (char *)malloc(1);
            if (text)
                text[0] = '\0';
            n = 0;
        }
        *out_text = text;
        *out_len = text ? n : 0;
        if (!text)
            rc = HGX_ERR;
    } else {
        hgx::textFree(text);
    }
    return rc;
}

static ColumnOptions columnOptions(const hgx_column_opts *o) {
    ColumnOptions c;
    if (o) {
        c.noDupes = o->no_dupes != 0;
        c.noAncestors = o->no_ancestors != 0;
        c.onlyOrthologs = o->only_orthologs != 0;
        if (o->n_targets > 0 && o->targets)
            c.targets.assign(o->targets, o->targets + o->n_targets);
    }
    return c;
}

int hgx_columns_depth(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int count_dupes,
                      const hgx_column_opts *opts, int32_t *out, char **err) {
    HGX_TRY
    if (!h || (count > 0 && !out))
        throw std::runtime_error("hgx_columns_depth: null argument");
    columnsDepthHost(h, ref, first, count, step, count_dupes ? 1 : 0, columnOptions(opts), out, nullptr);
    return HGX_OK;
    HGX_CATCH
}

int hgx_columns_depth_device(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, int count_dupes,
                             const hgx_column_opts *opts, int32_t *d_out, void *hip_stream, double *kernel_ms, char **err) {
    HGX_TRY
    if (!h || (count > 0 && !d_out))
        throw std::runtime_error("hgx_columns_depth_device: null argument");
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    ColumnStats st;
    columnsDepthDevice(h, ref, first, count, step, count_dupes ? 1 : 0, columnOptions(opts), d_out, hip_stream, &st);
    if (kernel_ms)
        *kernel_ms = st.depth_ms;
    return HGX_OK;
    HGX_CATCH
}

int hgx_columns_depth_stats(hgx_alignment *h, int ref, int64_t first, int64_t count, int64_t step, const hgx_column_opts *opts,
                            uint64_t *top_derefs, uint64_t *bottom_derefs, char **err) {
    HGX_TRY
    if (!h)
        throw std::runtime_error("hgx_columns_depth_stats: null argument");
    if (!h->dev)
        throw std::runtime_error("alignment was opened without a device (device = -1); the column engine needs the HIP path");
    ColumnStats st;
    std::vector<int32_t> scratch((size_t)std::max<int64_t>(count, 0));
    columnsDepthHost(h, ref, first, count, step, 0, columnOptions(opts), scratch.data(), &st, true);
    if (top_derefs)
        *top_derefs = st.top_derefs;
    if (bottom_derefs)
        *bottom_derefs = st.bottom_derefs;
    return HGX_OK;
    HGX_CATCH
}

int hgx_column_rows(hgx_alignment *h, int ref, int64_t first, int64_t count, const hgx_column_opts *opts, uint64_t **row_offset,
                    hgx_column_row **rows, size_t *n_rows, char **err) {
    HGX_TRY
    static_assert(sizeof(hgx_column_row) == sizeof(ColumnRowHost), "row layouts must match");
    if (!h || !row_offset || !rows || !n_rows)
        throw std::runtime_error("hgx_column_rows: null argument");
    if (!genomeOf(h, ref))
        throw std::runtime_error("hgx_column_rows: reference genome id out of range");
    if (count < 0 || first < 0)
        throw std::runtime_error("hgx_column_rows: negative column range");
    std::vector<uint64_t> off;
    std::vector<ColumnRowHost> r;
    columnsRowsHost(h, ref, first, count, columnOptions(opts), !h->img.genomes[(size_t)ref].dna.empty(), off, r, nullptr);
    *row_offset = (uint64_t *)malloc(off.size() * 8);
    *rows = (hgx_column_row *)malloc(std::max<size_t>(1, r.size()) * sizeof(hgx_column_row));
    if (!*row_offset || !*rows)
        throw std::runtime_error("out of memory");
    memcpy(*row_offset, off.data(), off.size() * 8);
    if (!r.empty())
        memcpy(*rows, r.data(), r.size() * sizeof(hgx_column_row));
    *n_rows = r.size();
    return HGX_OK;
    HGX_CATCH
}

// The text the export entry points hand back: an ostream over a malloc'd buffer that grows by realloc (large blocks move by
// remapping, not copying) and is handed to the caller as it stands.  (std::ostringstream + str() + a copy into malloc'd memory
// moved the 264 MB of an 8 M-column MAF four times.)
namespace {
class MallocBuf : public std::streambuf, public hgx::BulkSink {
  public:
    char *room(size_t n) override { // (hgx::BulkSink)
        if (!reserve(_n + n + 1))
            return nullptr;
        char *p = _p + _n;
        _n += n;
        return p;
    }
    ~MallocBuf() override { hgx::textFree(_p); }
    // NUL-terminated; the caller owns it (hgx_free)
    bool release(char **out, size_t *len) {
        if (!reserve(_n + 1))
            return false;
        _p[_n] = 0;
        *out = _p;
        *len = _n;
        _p = nullptr;
        _n = _cap = 0;
        return true;
    }

  protected:
    std::streamsize xsputn(const char *s, std::streamsize n) override {
        if (n <= 0)
            return 0;
        if (!reserve(_n + (size_t)n + 1))
            return 0;
        memcpy(_p + _n, s, (size_t)n);
        _n += (size_t)n;
        return n;
    }
    // (tellp: MafExport writes its header when the stream is at position 0, halMafExport.cpp:147-156)
    pos_type seekoff(off_type off, std::ios_base::seekdir dir, std::ios_base::openmode which) override {
        if (off == 0 && dir == std::ios_base::cur && (which & std::ios_base::out))
            return pos_type((off_type)_n);
        return pos_type(off_type(-1));
    }
    int_type overflow(int_type c) override {
        if (c == traits_type::eof())
            return traits_type::not_eof(c);
        const char ch = traits_type::to_char_type(c);
        return xsputn(&ch, 1) == 1 ? c : traits_type::eof();
    }

  private:
    bool reserve(size_t need) {
        if (need <= _cap)
            return true;
        size_t cap = _cap ? _cap : (size_t)1 << 16;
        while (cap < need)
            cap += cap < ((size_t)1 << 28) ? cap : ((size_t)1 << 28);
        // (hgx_textmem.hpp: a text of a megabyte or more lives in a mapping advised as huge pages and grows by mremap)
        char *q;
        if (_p && !hgx::textOwns(_p) && cap >= ((size_t)1 << 20)) { // outgrows malloc: moved over
            q = (char *)hgx::textAlloc(cap);
            if (q) {
                memcpy(q, _p, _n);
                free(_p);
            }
        } else {
            q = (char *)hgx::textRealloc(_p, cap);
        }
        if (!q)
            return false;
        _p = q;
        _cap = cap;
        return true;
    }
    char *_p = nullptr;
    size_t _n = 0, _cap = 0;
};
struct TextOut {
    MallocBuf buf;
    std::ostream os{&buf};
    int finish(char **out_text, size_t *out_len) {
        os.flush();
        if (!os.good() || !buf.release(out_text, out_len))
            throw std::runtime_error("out of memory for the output text");
        return HGX_OK;
    }
};
} // namespace

int hgx_alignment_depth(hgx_alignment *h, int ref, int ref_sequence, int64_t start, int64_t length, int64_t step, int count_dupes,
                        int no_ancestors, const int32_t *targets, int32_t n_targets, char **out_text, size_t *out_len, char **err) {
    HGX_TRY
    if (!h || !out_text || !out_len)
        throw std::runtime_error("hgx_alignment_depth: null argument");
    const GenomeTables *G = genomeOf(h, ref);
    if (!G || ref_sequence >= (int)G->seqs.size())
        throw std::runtime_error("hgx_alignment_depth: genome or sequence out of range");
    if (!G->children.empty() && no_ancestors) // halAlignmentDepth.cpp:182-187
        throw std::runtime_error("--noAncestors cannot be used when reference genome (" + G->name + ") is ancetral");
    std::set<int> tset(targets, targets + (targets ? n_targets : 0));
    TextOut T;
    std::ostream &os = T.os;
    alignmentDepth(os, h, ref, ref_sequence, tset, start, length, step, count_dupes != 0, no_ancestors != 0);
    return T.finish(out_text, out_len);
    HGX_CATCH
}

static void configureMaf(MafExport &me, const hgx_maf_opts *o, const GenomeTables *G);
// the caller's options as this build's struct: what the caller's header did not have yet reads as 0 (hgx_maf_opts.struct_size)
static hgx_maf_opts mafOpts(const hgx_maf_opts *o) {
    hgx_maf_opts x;
    memset(&x, 0, sizeof x);
    x.struct_size = (uint32_t)sizeof x;
    if (!o)
        return x;
    const size_t first = offsetof(hgx_maf_opts, max_block_len) + sizeof(int64_t);
    if (o->struct_size < first)
        throw std::runtime_error("hgx_maf_opts.struct_size is not set (initialise the options with HGX_MAF_OPTS_INIT)");
    if (o->struct_size > 4096 || o->struct_size % 4 != 0) // (a struct of a header to come is larger, not this large: garbage, or a caller built against hgx 0.1)
        throw std::runtime_error("hgx_maf_opts.struct_size is implausible (" + std::to_string(o->struct_size) +
                                 "): initialise the options with HGX_MAF_OPTS_INIT of this library's include/hgx.h");
    memcpy(&x, o, std::min<size_t>(o->struct_size, sizeof x));
    x.struct_size = (uint32_t)sizeof x;
    return x;
}

int hgx_maf_export_global(hgx_alignment *h, const hgx_maf_opts *o_, char **out_text, size_t *out_len, char **err) {
    HGX_TRY
    if (!h || !out_text || !out_len)
        throw std::runtime_error("hgx_maf_export_global: null argument");
    MafExport me;
    if (o_) {
        const hgx_maf_opts opts = mafOpts(o_), *o = &opts;
        me.setNoDupes(o->no_dupes != 0);
        me.setNoAncestors(o->no_ancestors != 0);
        me.setUcscNames(o->only_sequence_names == 0);
        me.setOnlyOrthologs(o->only_orthologs != 0);
        me.setMaxBlockLength(o->max_block_len == 0 ? 1000 : o->max_block_len);
        me.setPrintTree(o->print_tree != 0);
    }
    TextOut T;
    std::ostream &os = T.os;
    me.convertEntireAlignment(os, h);
    return T.finish(out_text, out_len);
    HGX_CATCH
}

// clones of handles[0] (the same host image), as hgx_liftover_convert_multi asks for
static std::vector<hgx_alignment *> cloneList(hgx_alignment *const *handles, int n_handles, const char *who) {
    if (!handles || n_handles < 1 || !handles[0])
        throw std::runtime_error(std::string(who) + ": no handles");
    std::vector<hgx_alignment *> hs;
    for (int i = 0; i < n_handles; ++i) {
        if (!handles[i] || handles[i]->imgHolder.get() != handles[0]->imgHolder.get())
            throw std::runtime_error(std::string(who) + ": every handle must be a device clone of the first (hgx_clone_to_device)");
        hs.push_back(handles[i]);
    }
    return hs;
}

int hgx_alignment_depth_multi(hgx_alignment *const *handles, int n_handles, int ref, int ref_sequence, int64_t start, int64_t length,
                              int64_t step, int count_dupes, int no_ancestors, const int32_t *targets, int32_t n_targets, char **out_text,
                              size_t *out_len, char **err) {
    HGX_TRY
    if (!out_text || !out_len)
        throw std::runtime_error("hgx_alignment_depth_multi: null argument");
    std::vector<hgx_alignment *> hs = cloneList(handles, n_handles, "hgx_alignment_depth_multi");
    hgx_alignment *h = hs[0];
    const GenomeTables *G = genomeOf(h, ref);
    if (!G || ref_sequence >= (int)G->seqs.size())
        throw std::runtime_error("hgx_alignment_depth_multi: genome or sequence out of range");
    if (!G->children.empty() && no_ancestors)
        throw std::runtime_error("--noAncestors cannot be used when reference genome (" + G->name + ") is ancetral");
    std::set<int> tset(targets, targets + (targets ? n_targets : 0));
    std::vector<hgx_alignment *> more(hs.begin() + 1, hs.end());
    TextOut T;
    std::ostream &os = T.os;
    alignmentDepth(os, h, ref, ref_sequence, tset, start, length, step, count_dupes != 0, no_ancestors != 0, nullptr, &more);
    return T.finish(out_text, out_len);
    HGX_CATCH
}

int hgx_maf_export_multi(hgx_alignment *const *handles, int n_handles, int ref, int ref_sequence, int64_t start, int64_t length,
                         int64_t slice_size, const hgx_maf_opts *o_, const int32_t *targets, int32_t n_targets, char **out_text,
                         size_t *out_len, char **err) {
    HGX_TRY
    if (!out_text || !out_len)
        throw std::runtime_error("hgx_maf_export_multi: null argument");
    std::vector<hgx_alignment *> hs = cloneList(handles, n_handles, "hgx_maf_export_multi");
    const GenomeTables *G = genomeOf(hs[0], ref);
    if (!G || ref_sequence >= (int)G->seqs.size())
        throw std::runtime_error("hgx_maf_export_multi: genome or sequence out of range");
    MafExportSettings cfg;
    if (o_) {
        const hgx_maf_opts opts = mafOpts(o_), *o = &opts;
        cfg.noDupes = o->no_dupes != 0;
        cfg.noAncestors = o->no_ancestors != 0;
        cfg.ucscNames = o->only_sequence_names == 0;
        cfg.onlyOrthologs = o->only_orthologs != 0;
        cfg.keepEmptyRefBlocks = o->keep_empty_ref_blocks != 0;
        cfg.unique = o->unique != 0;
        cfg.maxBlockLength = o->max_block_len == 0 ? 1000 : o->max_block_len;
        cfg.maxRefGap = o->max_ref_gap < 0 ? 0 : o->max_ref_gap;
        cfg.printTree = o->print_tree != 0;
        if (o->no_ancestors && !G->children.empty())
            throw std::runtime_error("Since the reference genome to be used for the MAF is ancestral (" + G->name +
                                     "), the --noAncestors option is invalid.  The --refGenome option can be used to specify a "
                                     "different reference.");
    }
    std::set<int> tset(targets, targets + (targets ? n_targets : 0));
    TextOut T;
    std::ostream &os = T.os;
    mafExportSliced(os, hs, ref, ref_sequence, start, length, slice_size, cfg, tset);
    return T.finish(out_text, out_len);
    HGX_CATCH
}

int hgx_maf_export_bed(hgx_alignment *h, int ref, const char *bed_text, size_t bed_len, const hgx_maf_opts *o, const int32_t *targets,
                       int32_t n_targets, char **out_text, size_t *out_len, char **err) {
    HGX_TRY
    if (!h || !out_text || !out_len || (bed_len && !bed_text))
        throw std::runtime_error("hgx_maf_export_bed: null argument");
    const GenomeTables *G = genomeOf(h, ref);
    if (!G)
        throw std::runtime_error("hgx_maf_export_bed: genome out of range");
    MafExport me;
    configureMaf(me, o, G);
    std::set<int> tset(targets, targets + (targets ? n_targets : 0));
    std::istringstream is(std::string(bed_text ? bed_text : "", bed_len));
    TextOut T;
    std::ostream &os = T.os;
    me.convertBed(os, h, ref, is, tset);
    return T.finish(out_text, out_len);
    HGX_CATCH
}

static void configureMaf(MafExport &me, const hgx_maf_opts *o_, const GenomeTables *G) {
    if (o_) {
        const hgx_maf_opts opts = mafOpts(o_), *o = &opts;
        me.setNoDupes(o->no_dupes != 0);
        me.setNoAncestors(o->no_ancestors != 0);
        me.setUcscNames(o->only_sequence_names == 0);
        me.setOnlyOrthologs(o->only_orthologs != 0);
        me.setKeepEmptyRefBlocks(o->keep_empty_ref_blocks != 0);
        me.setUnique(o->unique != 0);
        me.setMaxBlockLength(o->max_block_len == 0 ? 1000 : o->max_block_len);
        me.setMaxRefGap(o->max_ref_gap < 0 ? 0 : o->max_ref_gap);
        me.setPrintTree(o->print_tree != 0);
        if (o->no_ancestors && !G->children.empty()) // hal2maf.cpp:153-159
            throw std::runtime_error("Since the reference genome to be used for the MAF is ancestral (" + G->name +
                                     "), the --noAncestors option is invalid.  The --refGenome option can be used to specify a "
                                     "different reference.");
    }
}

int hgx_maf_export(hgx_alignment *h, int ref, int ref_sequence, int64_t start, int64_t length, const hgx_maf_opts *o,
                   const int32_t *targets, int32_t n_targets, char **out_text, size_t *out_len, char **err) {
    HGX_TRY
    if (!h || !out_text || !out_len)
        throw std::runtime_error("hgx_maf_export: null argument");
    const GenomeTables *G = genomeOf(h, ref);
    if (!G || ref_sequence >= (int)G->seqs.size())
        throw std::runtime_error("hgx_maf_export: genome or sequence out of range");
    MafExport me;
    configureMaf(me, o, G);
    std::set<int> tset(targets, targets + (targets ? n_targets : 0));
    TextOut T;
    std::ostream &os = T.os;
    if (ref_sequence >= 0) {
        me.convertSequence(os, h, ref, ref_sequence, start, length, tset);
    } else {
        for (size_t s = 0; s < G->seqs.size(); ++s)
            me.convertSequence(os, h, ref, (int)s, start, length, tset);
    }
    return T.finish(out_text, out_len);
    HGX_CATCH
}

int hgx_rand_preset(const char *preset, hgx_rand_opts *o) {
    if (!preset || !o)
        return HGX_ERR;
    RandOptions r;
    r.seed = o->seed;
    r.withDna = o->with_dna == 2 ? 2 : (o->with_dna != 0 ? 1 : 0);
    if (!randPreset(preset, r))
        return HGX_ERR;
    o->mean_degree = r.meanDegree;
    o->max_branch_length = r.maxBranchLength;
    o->min_genomes = r.minGenomes;
    o->max_genomes = r.maxGenomes;
    o->min_segment_length = r.minSegmentLength;
    o->max_segment_length = r.maxSegmentLength;
    o->min_segments = r.minSegments;
    o->max_segments = r.maxSegments;
    return HGX_OK;
}

int hgx_create_random(const hgx_rand_opts *o, int device, hgx_alignment **out, char **err) {
    HGX_TRY
    if (!o || !out)
        throw std::runtime_error("hgx_create_random: null argument");
    RandOptions r;
    r.meanDegree = o->mean_degree;
    r.maxBranchLength = o->max_branch_length;
    r.minGenomes = o->min_genomes;
    r.maxGenomes = o->max_genomes;
    r.minSegmentLength = o->min_segment_length;
    r.maxSegmentLength = o->max_segment_length;
    r.minSegments = o->min_segments;
    r.maxSegments = o->max_segments;
    r.seed = o->seed;
    r.withDna = o->with_dna == 2 ? 2 : (o->with_dna != 0 ? 1 : 0);
    return finishHandle(createRandomAlignment(r), device, out);
    HGX_CATCH
}

int hgx_save_image(const hgx_alignment *h, const char *path, char **err) {
    HGX_TRY
    if (!h || !path)
        throw std::runtime_error("hgx_save_image: null argument");
    writeImage(h->img, path);
    return HGX_OK;
    HGX_CATCH
}

void hgx_free(void *p) {
    hgx::textFree(p); // (the library's large texts are mappings: hgx_textmem.hpp; everything else is malloc's)
}

int hgx_maf_tracks_info(hgx_alignment *h, int drop, char **json, char **err) {
    HGX_TRY
    if (!h || !json)
        throw std::runtime_error("hgx_maf_tracks_info: null argument");
    std::string s = h->dev ? mafTracksInfo(h) : std::string("{\"tracks\": false}");
    s.insert(s.size() - 1, ", \"last_export\": " + mafLastExportInfo()); // (the host side of the last run-compressed export of this process)
    if (drop && h->dev)
        mafTracksDrop(h);
    *json = (char *)malloc(s.size() + 1);
    if (!*json)
        throw std::bad_alloc();
    memcpy(*json, s.c_str(), s.size() + 1);
    return HGX_OK;
    HGX_CATCH
}

void hgx_release_cached(void) {
    hgx::textTrim();
    hgx::columnsReleaseCached();
}

const char *hgx_version(void) {
    return "hgx 0.2 (HAL API 2.2 semantics; gfx950)";
}

} // extern "C"
