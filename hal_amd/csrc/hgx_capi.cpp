// extern "C" boundary of libhgx (include/hgx.h).  No exception crosses it.
#include "../../include/hgx.h"
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
    int64_t pos = 0, ti = 0, bi = 0;
    for (int64_t s = 0; s < num_sequences; ++s) {
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
    return o ? *o : d;
}

int hgx_liftover_batch(hgx_alignment *h, int src, int tgt, size_t n, const hgx_interval *iv, const hgx_liftover_opts *opts,
                       hgx_record **out, size_t *n_out, char **err) {
    HGX_TRY
    if (!h || !out || !n_out || (n && !iv))
        throw std::runtime_error("hgx_liftover_batch: null argument");
    if (!genomeOf(h, src) || !genomeOf(h, tgt))
        throw std::runtime_error("hgx_liftover_batch: genome id out of range");
    std::vector<hgx_record> recs;
    liftoverBatchHost(h, src, tgt, n, iv, defaultOpts(opts), recs, nullptr);
    *out = (hgx_record *)malloc(std::max<size_t>(1, recs.size()) * sizeof(hgx_record));
    if (!*out)
        throw std::runtime_error("out of memory");
    if (!recs.empty())
        memcpy(*out, recs.data(), recs.size() * sizeof(hgx_record));
    *n_out = recs.size();
    return HGX_OK;
    HGX_CATCH
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

int hgx_liftover_last_stats(const hgx_liftover_plan *p, hgx_liftover_stats *out) {
    if (!p || !out)
        return HGX_ERR;
    *out = liftoverPlanStats(p);
    return HGX_OK;
}

int hgx_liftover_kernel_times(const hgx_liftover_plan *p, char **json) {
    if (!p || !json)
        return HGX_ERR;
    const std::string s = liftoverPlanKernelTimes(p);
    *json = (char *)malloc(s.size() + 1);
    if (!*json)
        return HGX_ERR;
    memcpy(*json, s.c_str(), s.size() + 1);
    return HGX_OK;
}

int hgx_liftover_convert(hgx_alignment *h, int src, const char *bed_text, size_t bed_len, int tgt, int bed_type, int traverse_dupes,
                         int out_psl, int out_psl_with_name, int coalescence_limit, char **out_text, size_t *out_len, char **err) {
    // Output produced before a failing input line is still returned (the reference has already written it
    // to the stream when it throws, halBedScanner.cpp:49-58).
    std::ostringstream os;
    int rc = HGX_OK;
    try {
        if (!h || !out_text || !out_len || (bed_len && !bed_text))
            throw std::runtime_error("hgx_liftover_convert: null argument");
        if (!genomeOf(h, src) || !genomeOf(h, tgt))
            throw std::runtime_error("hgx_liftover_convert: genome id out of range");
        std::istringstream is(std::string(bed_text ? bed_text : "", bed_len));
        Liftover lo;
        lo.convert(h, src, &is, tgt, &os, bed_type, traverse_dupes != 0, out_psl != 0, out_psl_with_name != 0, coalescence_limit);
    } catch (std::exception &e) {
        setErr(err, e.what());
        rc = HGX_ERR;
    } catch (...) {
        setErr(err, "unknown error");
        rc = HGX_ERR;
    }
    if (out_text && out_len) {
        const std::string s = os.str();
        *out_text = (char *)malloc(s.size() + 1);
        if (*out_text) {
            memcpy(*out_text, s.c_str(), s.size() + 1);
            *out_len = s.size();
        } else {
            *out_len = 0;
            rc = HGX_ERR;
        }
    }
    return rc;
}

int hgx_rand_preset(const char *preset, hgx_rand_opts *o) {
    if (!preset || !o)
        return HGX_ERR;
    RandOptions r;
    r.seed = o->seed;
    r.withDna = o->with_dna != 0;
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
    r.withDna = o->with_dna != 0;
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
    free(p);
}

const char *hgx_version(void) {
    return "hgx 0.1 (HAL API 2.2 semantics; gfx950)";
}

} // extern "C"
