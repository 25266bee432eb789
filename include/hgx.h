/* hgx.h — C ABI of libhgx: MI355X-native replacement for the traversal hot path of HAL
 * (halLiftover's block mapper; hal2maf / halAlignmentDepth's column engine).
 *
 * The reference (ComparativeGenomicsToolkit/hal, paths below are relative to its root) has no FFI on
 * this path: tools link libHal.a + libHalLiftover.a and call C++ classes.  Each entry point here names
 * the C++ interface it stands behind; conventions follow the one C API the reference does ship,
 * blockViz/inc/halBlockViz.h:134-225: int status (0 = ok, <0 = error), message returned through
 * `char **err` (malloc'd, release with hgx_free; pass NULL to ignore), results allocated by the
 * library and released with hgx_free.  No exceptions cross this boundary.  A handle may be used from
 * one host thread at a time (same as the reference's stateful Liftover object).
 */
#ifndef HGX_H
#define HGX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HGX_OK 0
#define HGX_ERR (-1)
#define HGX_NULL_INDEX (-1) /* api/impl/halCommon.cpp:18 */

typedef struct hgx_alignment hgx_alignment;

/* ---- opening an alignment: replaces openHalAlignment (api/impl/halAlignmentInstance.cpp:133-165) ----
 * path: mmap-format HAL file (api/mmap_impl; versions 1.0/1.1; dirty files refused like
 * mmapFile.cpp:96-98) or an HGX flat image.  device >= 0: HIP device ordinal the segment tables are
 * uploaded to (kernels then run there); device == -1: host tables only (metadata queries; any compute
 * entry point fails). */
int hgx_open(const char *path, int device, hgx_alignment **out, char **err);
void hgx_close(hgx_alignment *h);
/* A second handle of the same alignment with its tables on another device of this process (multi-GPU, one process): the
 * host image is shared, everything device-side (tables, plans, staging) belongs to the new handle.  Released with hgx_close
 * like any handle, in any order. */
int hgx_clone_to_device(const hgx_alignment *h, int device, hgx_alignment **out, char **err);

/* Build an alignment from caller-owned flat arrays instead of a file (the route for HDF5-backed
 * alignments: the maintainer-side loop over Genome::getTopSegmentIterator/getBottomSegmentIterator is
 * shown in INTEGRATION.md).  Usage: begin, add every genome parent-before-child (children of one parent
 * in child-index order, api/inc/halGenome.h:247-256), finish.
 *   top arrays have num_top entries, top_start additionally a sentinel (= genome length);
 *   child_index/child_reversed are [num_children][num_bottom], row-major. */
typedef struct hgx_builder hgx_builder;
int hgx_builder_begin(hgx_builder **out, char **err);
int hgx_builder_add_genome(hgx_builder *b, const char *name, const char *parent_name /* NULL for root */,
                           double branch_length, int64_t num_sequences, const char *const *seq_names,
                           const int64_t *seq_lengths, const int64_t *seq_num_top, const int64_t *seq_num_bottom,
                           int64_t num_top, const int64_t *top_start, const int64_t *top_parent_index,
                           const uint8_t *top_parent_reversed, const int64_t *top_next_paralogy,
                           const int64_t *top_bottom_parse, int64_t num_bottom, const int64_t *bottom_start,
                           const int64_t *bottom_top_parse, int64_t num_children, const int64_t *child_index,
                           const uint8_t *child_reversed, const char *dna /* may be NULL */, char **err);
int hgx_builder_finish(hgx_builder *b, int device, hgx_alignment **out, char **err); /* consumes b */
void hgx_builder_abort(hgx_builder *b);

/* ---- metadata: Alignment / Genome / Sequence getters (api/inc/halAlignment.h, halGenome.h) ---- */
int hgx_num_genomes(const hgx_alignment *h);
const char *hgx_newick(const hgx_alignment *h);                    /* Alignment::getNewickTree */
const char *hgx_genome_name(const hgx_alignment *h, int genome);   /* Genome::getName */
int hgx_genome_id(const hgx_alignment *h, const char *name);       /* Alignment::openGenome; -1 if absent */
int hgx_genome_parent(const hgx_alignment *h, int genome);         /* Genome::getParent; -1 for root */
int hgx_genome_num_children(const hgx_alignment *h, int genome);   /* Genome::getNumChildren */
int hgx_genome_child(const hgx_alignment *h, int genome, int k);   /* Genome::getChild */
int64_t hgx_genome_length(const hgx_alignment *h, int genome);     /* Genome::getSequenceLength */
int64_t hgx_genome_num_top(const hgx_alignment *h, int genome);    /* Genome::getNumTopSegments */
int64_t hgx_genome_num_bottom(const hgx_alignment *h, int genome); /* Genome::getNumBottomSegments */
int hgx_genome_num_sequences(const hgx_alignment *h, int genome);  /* Genome::getNumSequences */
/* Sequence::getName / getStartPosition / getSequenceLength by index */
int hgx_sequence_info(const hgx_alignment *h, int genome, int seq, const char **name, int64_t *start, int64_t *length);
/* Genome::getSequence(name) (api/mmap_impl/mmapGenome.cpp:186-196); returns the sequence index or -1 */
int hgx_sequence_lookup(const hgx_alignment *h, int genome, const char *name, int64_t *start, int64_t *length);
/* getLowestCommonAncestor (api/impl/halCommon.cpp:123-152) of two genomes */
int hgx_mrca(const hgx_alignment *h, int genome_a, int genome_b);

/* ---- halLiftover hot path: BlockLiftover::liftInterval (liftover/impl/halBlockLiftover.cpp:46-113) =
 * halMapSegment per source segment (api/impl/halSegmentMapper.cpp:639-670) + insertAndBreakOverlaps
 * (:475-520) + BlockMapper::extractSegment (liftover/impl/halBlockMapper.cpp:331-394) + the stable sort by
 * source start of Liftover::visitLine (liftover/impl/halLiftover.cpp:90). ---- */
typedef struct hgx_interval {
    int64_t start; /* BED chromStart, sequence-relative, 0-based */
    int64_t end;   /* BED chromEnd, exclusive */
    int32_t seq;   /* sequence index in the source genome */
    char strand;   /* '+', '-' or '.' */
    char _pad[3];
} hgx_interval;

typedef struct hgx_record { /* one lifted output interval = one output BED line */
    int64_t query;     /* index of the input interval this line came from */
    int64_t tgt_start; /* sequence-relative, 0-based */
    int64_t tgt_end;   /* exclusive */
    int64_t src_start; /* genome coordinate in the source; the reference's hidden sort key (_srcStart) */
    int32_t tgt_seq;   /* sequence index in the target genome */
    char strand;       /* '+', '-' ('.' when the input strand was '.') */
    uint8_t tgt_reversed; /* orientation of the mapped piece itself (also set when strand is '.'); needed for PSL */
    char _pad[2];
} hgx_record;

typedef struct hgx_liftover_opts {
    int32_t traverse_dupes;    /* Liftover::convert traverseDupes (halLiftover.h:25-28); --noDupes => 0 */
    int32_t coalescence_limit; /* --coalescenceLimit: genome id of an ancestor of the MRCA, or -1 (= the MRCA, the default);
                                  paralogs coalescing up to that genome are followed (halSegmentMapper.cpp:525-576) */
    int64_t min_length;        /* halMapSegment minLength; halLiftover passes 0 */
    int32_t emit_blocks;       /* 2: as 1 below, but the pieces as halMapSegment's walk leaves them, before insertAndBreakOverlaps
                                  cuts them against each other (sorted in set order, equal ones once);
                                  1: records are the members of the mapped set itself, in set order (BlockMapper::getMap,
                                  liftover/inc/halBlockMapper.h:36) instead of merged output lines: tgt_start/tgt_end = the
                                  member's forward target range, src_start = its forward source start, strand = orientation
                                  of its source side ('+'/'-'), tgt_reversed = orientation of its target side */
    int32_t block_mapper_source; /* 1: walk the source genome the way BlockMapper::map does (halBlockMapper.cpp:79-86: bottom
                                    segments iff it is the MRCA and not the target) instead of BlockLiftover's rule;
                                    2 / 3: through its top / bottom tiling, whatever the rules say (what mapAdjacencies'
                                    iterator does: it stands on the tiling its mapped segment lies on, :247-256) */
} hgx_liftover_opts;

/* Host-buffer form.  Records come back grouped by input interval in input order and, inside one
 * interval, in the reference's print order.  Intervals with end > sequence length are skipped (no
 * records), like halLiftover.cpp:62-66.  *out is released with hgx_free. */
int hgx_liftover_batch(hgx_alignment *h, int src_genome, int tgt_genome, size_t n, const hgx_interval *intervals,
                       const hgx_liftover_opts *opts, hgx_record **out, size_t *n_out, char **err);

/* Liftover::convert over several handles of one alignment (hgx_clone_to_device), one per GPU: the input's lines are dealt to
 * the devices in contiguous shares, lifted at the same time, and the output is put together in input order on the host —
 * the one-process form of sharding independent intervals (liftover/impl/halLiftover.cpp:46-92 clears all state per line).
 * Same arguments, output and errors as hgx_liftover_convert; inputs the parallel text path does not take (BED12, PSL, mixed
 * column counts) run on handles[0] alone. */
int hgx_liftover_convert_multi(hgx_alignment *const *handles, int n_handles, int src_genome, const char *bed_text, size_t bed_len,
                               int tgt_genome, int bed_type, int traverse_dupes, int out_psl, int out_psl_with_name, int coalescence_limit,
                               char **out_text, size_t *out_len, char **err);

/* BlockMapper (liftover/inc/halBlockMapper.h:30-40) without adjacencies: init(refGenome, queryGenome, absRefFirst,
 * absRefLast, targetReversed, doDupes, minLength, false, coalescenceLimit); map(); getMap().  abs_ref_first/last are
 * genome coordinates, last inclusive.  Records as described at hgx_liftover_opts.emit_blocks, query = 0; the source
 * range of a record is [src_start, src_start + (tgt_end - tgt_start)).  Released with hgx_free. */
int hgx_block_map(hgx_alignment *h, int ref_genome, int query_genome, int64_t abs_ref_first, int64_t abs_ref_last,
                  int target_reversed, int do_dupes, int64_t min_length, int coalescence_limit, hgx_record **out, size_t *n_out,
                  char **err);

/* ---- the blockViz query: halGetBlocksInTargetRange (blockViz/inc/halBlockViz.h:222-225; impl/halBlockViz.cpp:243-330) =
 * BlockMapper::init + map WITH adjacencies (liftover/impl/halBlockMapper.cpp:36-245: for every mapped segment the stretch of the
 * query genome next to it on either side is mapped back to the target genome) + chainReferenceParalogies + extractSegment with the
 * paralogy set and the range's ends as cut points + readBlock + processTargetDupes (halBlockViz.cpp:759-1178).
 * The structs have the layout of hal_block_t / hal_target_range_t / hal_target_dupe_list_t / hal_block_results_t
 * (halBlockViz.h:23-58: int64_t where the reference has `long`), so a maintainer's halGetBlocksInTargetRange can return the
 * library's result as it is; every string and node is malloc'd, released by hgx_free_block_results.
 * q_species / t_species / t_chrom: query genome, target (reference) genome and its sequence; [t_start, t_end) sequence-relative,
 * t_end == 0: to the end of the sequence; t_reversed, seq_mode (0 none, otherwise DNA strings: there are no levels of detail
 * here), dup_mode (0 none, 1 query dupes, 2 query and target dupes), map_back_adjacencies and coalescence_limit_name (NULL: the
 * MRCA — the root for a self-alignment) as halGetBlocksInTargetRange's.  Returns NULL and *err (release with hgx_free) on failure,
 * with the reference's messages for its own argument checks. */
typedef struct hgx_target_range {
    struct hgx_target_range *next;
    int64_t tStart, size;
} hgx_target_range;
typedef struct hgx_target_dupe_list {
    struct hgx_target_dupe_list *next;
    int64_t id;
    hgx_target_range *tRange;
    char *qChrom;
} hgx_target_dupe_list;
typedef struct hgx_block { /* ALL COORDINATES ARE FORWARD-STRAND RELATIVE (halBlockViz.h:47-58) */
    struct hgx_block *next;
    char *qChrom;
    int64_t tStart, qStart, size;
    char strand;
    char *qSequence, *tSequence; /* NULL unless asked for */
} hgx_block;
typedef struct hgx_block_results {
    hgx_block *mappedBlocks;
    hgx_target_dupe_list *targetDupeBlocks;
} hgx_block_results;
hgx_block_results *hgx_get_blocks_in_target_range(hgx_alignment *h, const char *q_species, const char *t_species, const char *t_chrom,
                                                  int64_t t_start, int64_t t_end, int64_t t_reversed, int seq_mode, int dup_mode,
                                                  int map_back_adjacencies, const char *coalescence_limit_name, char **err);
/* The same for n ranges of one target sequence at once — one request per track window, or a genome browsed in tiles: the
 * mapping of all ranges is one batch of device work, and so is the mapping back of all their adjacencies; the sequential part of
 * BlockMapper::mapAdjacencies (every adjacency added changes what the next one is cut against) runs per range on the host.
 * results[k] (caller's array of n pointers) receives what hgx_get_blocks_in_target_range returns for [t_starts[k], t_ends[k]);
 * HGX_ERR with *err set and every results[k] NULL on failure. */
int hgx_get_blocks_in_target_ranges(hgx_alignment *h, const char *q_species, const char *t_species, const char *t_chrom, size_t n,
                                    const int64_t *t_starts, const int64_t *t_ends, int64_t t_reversed, int seq_mode, int dup_mode,
                                    int map_back_adjacencies, const char *coalescence_limit_name, hgx_block_results **results, char **err);
void hgx_free_block_results(hgx_block_results *results); /* halFreeBlockResults */

/* Device-resident form: the query table is already in HBM and the records stay there.
 * A plan owns the per-(src,tgt) walk schedule and all device workspaces, sized for max_queries. */
typedef struct hgx_liftover_plan hgx_liftover_plan;
int hgx_liftover_plan_create(hgx_alignment *h, int src_genome, int tgt_genome, const hgx_liftover_opts *opts,
                             size_t max_queries, hgx_liftover_plan **out, char **err);
void hgx_liftover_plan_destroy(hgx_liftover_plan *p);
/* d_gstart / d_gend: device int64[n], inclusive GENOME coordinates (start + sequence start, end - 1 +
 * sequence start: halBlockLiftover.cpp:48-49); d_strand: device uint8[n] ('+','-','.').
 * hip_stream: hipStream_t (NULL = default stream).  On return the run is complete and
 * *d_records (device, owned by the plan, valid until the next run) holds *n_records hgx_record. */
int hgx_liftover_run_device(hgx_liftover_plan *p, size_t n, const int64_t *d_gstart, const int64_t *d_gend,
                            const uint8_t *d_strand, void *hip_stream, const hgx_record **d_records,
                            size_t *n_records, char **err);
/* The same in two halves, for callers that keep several batches in flight: hgx_liftover_submit queues the batch's launches on
 * hip_stream and returns; hgx_liftover_collect waits for them and hands back what hgx_liftover_run_device hands back.  A plan
 * has one batch in flight (its workspaces and its output belong to that batch until the next submit); two plans of the same
 * alignment on two streams overlap one batch's last wavefronts and the host's launch and wake-up times with the other batch.
 * The interval arrays must stay as they are until collect returns.  (Liftover::convert's lines are independent,
 * liftover/impl/halLiftover.cpp:46-92: batches may be mapped in any order and at the same time.) */
int hgx_liftover_submit(hgx_liftover_plan *p, size_t n, const int64_t *d_gstart, const int64_t *d_gend, const uint8_t *d_strand,
                        void *hip_stream, char **err);
int hgx_liftover_collect(hgx_liftover_plan *p, const hgx_record **d_records, size_t *n_records, char **err);
/* Counters of the last run (for roofline accounting): queries, source pieces, top-segment records
 * dereferenced, bottom-segment records dereferenced, mapped pieces before merging, output records,
 * and the accumulated device time in ms of the walk kernels / all kernels of the run. */
typedef struct hgx_liftover_stats {
    uint64_t queries, source_pieces, top_derefs, bottom_derefs, mapped_pieces, records, deferred_queries;
    double walk_ms, total_ms;
    /* A plan whose intervals have reached a quarter of the source genome's top segments (the batch that crosses the line
     * included) is served from a composed table, built once per alignment, genome pair and options on the device (every
     * source top segment runs through the walk kernels as one interval, the pieces are radix-sorted by source position;
     * about the cost of walking one interval per source segment):
     *   composed_kind 3 — the MERGED table of the whole path: the pieces of kind 2 joined into the maximal chains that
     *                     canMergeRightWith would merge (rows of a chain file).  One launch counts every interval's
     *                     lines, one writes every record at its final place; intervals that may need overlap breaking (one
     *                     of their records overlaps on the target with a record nearby in the source; general_queries of
     *                     them) are finished from the kind-2 table inside the counting launch.  32-bit coordinates only;
     *                     HGX_MERGED=0 forbids it;
     *   composed_kind 2 — the table of the whole path source -> MRCA -> target (paralogy rings and coalescenceLimit
     *                     included): an interval is one lookup plus the grouping / merging step;
     *   composed_kind 1 — the up table source -> MRCA (when the target is the MRCA itself, or HGX_COMPOSED_THROUGH=0):
     *                     locate + up phase from the table, then the ordinary down phase;
     *   composed_kind 0 — this plan walks level by level.
     * composed_records / composed_build_ms: size and build time of the table.  With a table, top_derefs counts the table
     * records dereferenced plus the segment records of whatever part of the walk still runs.
     * HGX_COMPOSED_UP=1 in the environment builds the table when the plan is created, =0 forbids it;
     * HGX_COMPOSED_AFTER=x moves the change-over to x times the source's segments. */
    uint64_t composed_records;
    double composed_build_ms;
    uint64_t composed_kind;
    uint64_t general_queries;  /* kind 3: intervals of the last run that took the general route */
    uint64_t composed_flagged; /* kind 3: flagged records of the table */
} hgx_liftover_stats;
int hgx_liftover_last_stats(const hgx_liftover_plan *p, hgx_liftover_stats *out);
/* Per-kernel device time, measured with HIP events on the run's stream, as a JSON object
 * {"kernel": {"ms": total, "launches": n, "top_derefs": t, "bot_derefs": b}, ...}; release *json with hgx_free.
 * hgx_liftover_plan_set_timing chooses what it covers: 1 = the last run (default), 2 = every run since the previous
 * read (for a benchmark loop: the elapsed-time queries then happen once, outside the loop), 0 = no events at all. */
int hgx_liftover_kernel_times(hgx_liftover_plan *p, char **json);
int hgx_liftover_plan_set_timing(hgx_liftover_plan *p, int mode);
/* The general intervals of a single-pass batch (kind 3; BlockMapper's general algorithm, liftover/impl/halBlockMapper.cpp:331-394,
 * for the few intervals whose records overlap on the target) can be found in a pass of their own in front of the classifying
 * launch and finished by workgroups at the head of its grid, so that the launch does not end one such interval's latency after
 * its last tile.  n < 0 (default): done when a plan's last batch had few of them and the batch runs by itself
 * (hgx_liftover_run_device) — batches kept in flight (hgx_liftover_submit_device) overlap their launches' tails anyway and are
 * spared the extra pass over their intervals; 0: never; n > 0: always, with n workgroups.  Same records either way. */
int hgx_liftover_plan_set_workers(hgx_liftover_plan *p, int n);
/* Where a plan's change-over to its table spent its time: with HGX_BUILD_TIMING=2 in the environment (1: also printed to
 * stderr) the run that builds a table synchronises the device at every phase boundary and keeps the wall time of each phase —
 * builder plan, walk of every source segment, sorts, records, bucket tables, junction sort, chains, flags, the releases of the
 * workspaces, the first batch served from the table.  *json: the phases of this process's last such run as a JSON list of
 * [name, ms] (they add up to that run's wall time, a little more than an un-instrumented run's); release with hgx_free. */
int hgx_liftover_build_phases(char **json);

/* Copy the plan-owned records of the last run into caller-owned device memory (device to device, on hip_stream). */
int hgx_liftover_copy_records(const hgx_liftover_plan *p, void *d_dst, size_t n_records, void *hip_stream, char **err);
/* Same in the 20-byte wire form of the multi-GPU exchange: five int32 per record — query, tgt_start, tgt_end, src_start,
 * tgt_seq << 16 | strand << 8 | tgt_reversed.  Only meaningful when every coordinate and the batch size fit 31 bits and the
 * target genome has fewer than 65536 sequences (the caller checks: hal_amd.shard.can_pack). */
int hgx_liftover_copy_records_packed(const hgx_liftover_plan *p, void *d_dst, size_t n_records, void *hip_stream, char **err);

/* The last run's records as one self-describing blob for the all-gatherv of the multi-GPU path (hal_amd/shard.py decodes
 * it): a 32-byte header {"HGXW", uint32 format, int64 first_query, uint64 n_queries, uint64 n_records} and then
 *   format 12: uint16 record count per interval (padded to 8 bytes), then 12 bytes per record — tgt_start and src_start
 *              as uint32, (tgt_end - tgt_start) | tgt_seq << 22 | strand code << 29 | tgt_reversed << 31 (strand code
 *              0 '+', 1 '-', 2 '.'); the query index follows from the counts and first_query;
 *   format 8:  the same counts, then 8 bytes per record — tgt_start and the third word of format 12, no source coordinate:
 *              all a receiver needs that only prints BED lines (liftover/impl/halLiftover.cpp:94-106).  Written only when
 *              the caller sets *format to 8 before the call, under format 12's conditions (else 20 / 40 as below);
 *   format 20: the records of hgx_liftover_copy_records_packed (query relative to first_query);
 *   format 40: hgx_record rows as they are (query relative to first_query).
 * Format 12 is used when every field fits (lengths < 2^22, positions < 2^32, < 128 target sequences, < 65536 records per
 * interval), which the call checks on the device; otherwise format 20 when both genomes are shorter than 2^31 bases and
 * the target has at most 65536 sequences, else format 40.  With d_dst == NULL only *bytes is set, to the
 * capacity a destination needs; otherwise *bytes is what was written and *format which form.  first_query: global index of
 * the batch's first interval (what shard_bounds gave this rank). */
int hgx_liftover_wire_blob(hgx_liftover_plan *p, void *d_dst, size_t capacity, int64_t first_query, size_t *bytes, int *format,
                           void *hip_stream, char **err);

/* ---- the exchange step between the GPUs of a node, one process per GPU (RCCL, loaded by the library at run time) ----
 * hgx_comm_unique_id: made by one rank and handed to all (file, MPI, torch.distributed ...: 128 bytes);
 * hgx_comm_create: collective over the n_ranks ranks, each on its own device.
 * hgx_liftover_exchange: ONE collective per batch — this rank's records of the plan's last run travel as a wire blob
 * (hgx_liftover_wire_blob: self-describing, its header carries its sizes) in slot `rank` of d_gathered, a device buffer of
 * n_ranks slots of slot_bytes each; on return (stream-ordered on hip_stream) every slot holds the blob of its rank, and the
 * rank-major concatenation of their records is the unsharded output (intervals are independent: halLiftover.cpp:46-92).
 * A rank that has no blob for the batch still takes part and returns HGX_ERR: its slot's header has format 0, with n_queries 0
 * and the bytes it needed in n_records when its blob did not fit the slot, n_queries 1 when it failed for another reason (a
 * batch in flight, a HIP error).  The slot's header is cleared before the blob is built, so a slot never shows an earlier
 * batch's blob as this one's. */
typedef struct hgx_comm hgx_comm;
int hgx_comm_unique_id(unsigned char *id128, char **err);
int hgx_comm_create(const unsigned char *id128, int rank, int n_ranks, int device, hgx_comm **out, char **err);
void hgx_comm_destroy(hgx_comm *c);
int hgx_liftover_exchange(hgx_liftover_plan *p, hgx_comm *c, int64_t first_query, void *d_gathered, size_t slot_bytes, void *hip_stream,
                          size_t *my_bytes, char **err);
/* The collation a writer needs: every rank's blob in rank `root`'s buffer only (ncclSend / ncclRecv in one group; a rank sends
 * its slot once instead of receiving every other rank's — with N ranks an all-gather moves N times the batch's records into
 * every GPU, this moves them once into one).  d_gathered: n_ranks slots on the root, one slot (the rank's own, slot 0) on the
 * others.  bed_only: the blobs in the 8-byte form when they fit it (hgx_liftover_wire_blob).  Failures as
 * hgx_liftover_exchange: the rank still takes part. */
int hgx_liftover_gather(hgx_liftover_plan *p, hgx_comm *c, int root, int64_t first_query, void *d_gathered, size_t slot_bytes,
                        int bed_only, void *hip_stream, size_t *my_bytes, char **err);
/* Several writers instead of one root — the collation hal2mafMP.py / a pool of halLiftover processes do through files
 * (maf/hal2mafMP.py:176-190 concatenates the workers' outputs): the ranks are taken in groups of group_size consecutive ranks,
 * the first rank of a group is its writer and receives the group's blobs (d_gathered: group_size slots on a writer, the slot of
 * rank r at r - writer; one slot on the others).  A GPU's xGMI links are point to point, so the group_size - 1 blobs a writer
 * receives arrive over as many links at once — the step takes one blob's time over one link whatever the group and the world; what
 * the writers divide among themselves is the HOST work behind it (rendering and writing the lines of n_ranks / n_writers ranks
 * each).  Level two moves no records: hgx_comm_all_sizes tells every writer the bytes of text the writers before it hold, i.e.
 * where in the one output file its text belongs (rank-major order of the groups = input order), and the writers write side by
 * side (pwrite).  All ranks of the communicator call both; failures as hgx_liftover_exchange. */
int hgx_liftover_gather_writers(hgx_liftover_plan *p, hgx_comm *c, int group_size, int64_t first_query, void *d_gathered,
                                size_t slot_bytes, int bed_only, void *hip_stream, size_t *my_bytes, char **err);
/* every rank's `mine` in rank order on every rank (an all-gather of eight bytes a rank; blocking; sizes: n_ranks values on the host) */
int hgx_comm_all_sizes(hgx_comm *c, uint64_t mine, uint64_t *sizes, char **err);
/* What a writer does with its group's blobs once they are in host memory: the lifted BED text of the intervals they hold — the
 * lines BedLine::write prints (liftover/impl/halBedLine.cpp:104-151) with what BlockLiftover::liftInterval and
 * Liftover::cleanResults put into them, exactly as hgx_liftover_convert renders them.  src_bed: the input lines of the group's
 * ranks, in order (blob i answers the next n_queries(i) lines that are intervals of the source genome; lines of sequences it does
 * not have are passed over, as in the conversion); blobs[i] / blob_bytes[i]: slot i as hgx_liftover_gather_writers /
 * hgx_liftover_gather / hgx_liftover_exchange left it (any wire format), copied to the host.  BED3..BED9 lines (bed_type 0 = auto);
 * no device is needed.  *out_text is released with hgx_free. */
int hgx_liftover_render_blobs(hgx_alignment *h, int src_genome, int tgt_genome, const char *src_bed, size_t src_len, int bed_type,
                              const void *const *blobs, const size_t *blob_bytes, int n_blobs, char **out_text, size_t *out_len, char **err);


/* Text-level drop-in for Liftover::convert (liftover/inc/halLiftover.h:25-28): BED text in, BED text
 * out, byte-identical to halLiftover for BED3..BED9 (+ extra columns).  bed_type 0 = auto
 * (halBedLine.cpp:36-38).  *out_text is released with hgx_free. */
int hgx_liftover_convert(hgx_alignment *h, int src_genome, const char *bed_text, size_t bed_len, int tgt_genome,
                         int bed_type, int traverse_dupes, int out_psl, int out_psl_with_name, int coalescence_limit,
                         char **out_text, size_t *out_len, char **err);

/* ---- column engine: ColumnIterator (api/impl/halColumnIterator.cpp) in its default configuration
 * (maxInsertLength 0, unique false — columns are independent, :785-787), as used by halAlignmentDepth
 * (alignmentDepth/halAlignmentDepth.cpp:215-308) and hal2maf (maf/impl/halMafExport.cpp:25-88). ---- */
typedef struct hgx_column_opts {
    int32_t no_dupes;       /* ColumnIterator noDupes */
    int32_t no_ancestors;   /* ColumnIterator noAncestors */
    int32_t only_orthologs; /* ColumnIterator onlyOrthologs */
    int32_t n_targets;      /* 0 = visit everything */
    const int32_t *targets; /* genome ids (the reference genome is added, halColumnIterator.cpp:47) */
} hgx_column_opts;

/* One value per column for columns first, first+step, ... (count of them; `first` is a GENOME coordinate of
 * ref_genome): count_dupes == 0: genomes with a base in the column - 1; 1: bases in the column - 1
 * (halAlignmentDepth.cpp:258-281).  out: host int32[count]. */
int hgx_columns_depth(hgx_alignment *h, int ref_genome, int64_t first, int64_t count, int64_t step, int count_dupes,
                      const hgx_column_opts *opts, int32_t *out, char **err);
/* Same with the result left in HBM (d_out: device int32[count]); kernel_ms (may be NULL) receives the kernel's
 * device time measured with HIP events on `hip_stream`. */
int hgx_columns_depth_device(hgx_alignment *h, int ref_genome, int64_t first, int64_t count, int64_t step, int count_dupes,
                             const hgx_column_opts *opts, int32_t *d_out, void *hip_stream, double *kernel_ms, char **err);
/* Roofline accounting of the same computation: the segment records the column walks logically dereference (one per
 * top / bottom segment looked at; 25 bytes each).  Runs the kernel's counting instantiation (about 10 %
 * slower than the plain one, which is why the counts are not a by-product of hgx_columns_depth_device). */
int hgx_columns_depth_stats(hgx_alignment *h, int ref_genome, int64_t first, int64_t count, int64_t step, const hgx_column_opts *opts,
                            uint64_t *top_derefs, uint64_t *bottom_derefs, char **err);

/* Every reported base of columns [first, first+count) in ColumnMap insertion order: what a loop over
 * Sequence::getColumnIterator()/toRight()/getColumnMap() sees.  *row_offset has count+1 entries. */
typedef struct hgx_column_row {
    int64_t pos;    /* genome coordinate */
    int32_t genome;
    uint8_t reversed;
    char base;      /* DnaIterator::getBase (complemented when reversed); 'N' when the alignment has no DNA */
    uint8_t _pad[2];
} hgx_column_row;
int hgx_column_rows(hgx_alignment *h, int ref_genome, int64_t first, int64_t count, const hgx_column_opts *opts,
                    uint64_t **row_offset, hgx_column_row **rows, size_t *n_rows, char **err);

/* halAlignmentDepth's printGenome (alignmentDepth/halAlignmentDepth.cpp:318-347): wig text.
 * ref_sequence: sequence index or -1 for the whole genome; start/length as --start/--length (0 = to the end). */
int hgx_alignment_depth(hgx_alignment *h, int ref_genome, int ref_sequence, int64_t start, int64_t length, int64_t step,
                        int count_dupes, int no_ancestors, const int32_t *targets, int32_t n_targets, char **out_text,
                        size_t *out_len, char **err);

/* hal2maf: MafExport::convertSequence (maf/impl/halMafExport.cpp:25-88) over one sequence, or over every
 * sequence of the reference genome when ref_sequence == -1 (maf/impl/hal2maf.cpp:196-206).  MAF text, header
 * included, byte-identical to hal2maf for the options below. */
typedef struct hgx_maf_opts {
    uint32_t struct_size; /* sizeof(hgx_maf_opts) as the caller's header has it (HGX_MAF_OPTS_INIT sets it): fields the caller's
                             struct does not reach read as 0, a size below the first version's (up to max_block_len) is an error.
                             New options are added at the end only. */
    int32_t no_dupes, no_ancestors, only_sequence_names, only_orthologs, keep_empty_ref_blocks;
    int32_t unique; /* --unique: a column is written once, by its left-most reference base (halColumnIterator.cpp:210-214) */
    int64_t max_block_len; /* --maxBlockLen; 0 = the default, 1000 (halMafBlock.cpp:16).  A negative value is what the
                              reference makes of --maxBlockLen 0 (MafBlock::setMaxLength keeps the value, halMafBlock.h:133):
                              every column starts a block; the hal2maf twin passes the option through unchanged */
    int64_t max_ref_gap; /* --maxRefGap = ColumnIterator's maxInsertLength (maf/impl/halMafExport.cpp:47): > 0: between two
                            reference columns the bases the reference lacks — ranges deleted above it, inserted below it, up to
                            this many bases — come as columns of their own (halColumnIterator.cpp:65-144, 357-405), and, as in
                            the reference, every reference base is then written once (the visit cache is on) */
    int32_t print_tree;  /* --printTree: every block begins "a tree=\"...\"" with the tree of its rows, the rows in the tree's post
                            order; a block also ends where the column's tree changes (halMafBlock.cpp:121-292, 443-448, 485-497).
                            As in the reference, a column whose first base (in sequence order) is an insertion in a genome that
                            has bottom segments has no such tree: undefined behaviour there, an error here */
} hgx_maf_opts;
#define HGX_MAF_OPTS_INIT {(uint32_t)sizeof(hgx_maf_opts), 0, 0, 0, 0, 0, 0, 0, 0, 0}
int hgx_maf_export(hgx_alignment *h, int ref_genome, int ref_sequence, int64_t start, int64_t length, const hgx_maf_opts *opts,
                   const int32_t *targets, int32_t n_targets, char **out_text, size_t *out_len, char **err);
/* hal2maf --refTargets: one convertSequence per BED interval (or BED12 block) of the reference genome, sharing one
 * MafExport (MafBed::visitLine, maf/impl/halMafBed.cpp:24-52). */
int hgx_maf_export_bed(hgx_alignment *h, int ref_genome, const char *bed_text, size_t bed_len, const hgx_maf_opts *opts,
                       const int32_t *targets, int32_t n_targets, char **out_text, size_t *out_len, char **err);

/* hal2maf --global: MafExport::convertEntireAlignment (maf/impl/halMafExport.cpp:90-153) — every column of the alignment once, the
 * leaves taken as the reference one after the other, each with the visit cache of the ones before it.  Of opts the fields that
 * function reads: no_dupes, no_ancestors, only_sequence_names, only_orthologs, max_block_len. */
int hgx_maf_export_global(hgx_alignment *h, const hgx_maf_opts *opts, char **out_text, size_t *out_len, char **err);

/* The column tools over several handles of one alignment (hgx_clone_to_device), one per GPU of this process — columns are
 * independent (api/impl/halColumnIterator.cpp:785-787), so contiguous ranges of the reference scan at the same time:
 * hgx_alignment_depth_multi: the sampled columns of every sequence in contiguous shares, one per device; the wig text is the
 *   one hgx_alignment_depth writes (BASELINE config 5: the whole-genome depth scan on 8 GPUs);
 * hgx_maf_export_multi: the reference's own way of running hal2maf in parallel (maf/hal2mafMP.py:63-79, 176-190): every
 *   sequence's range cut into slices of slice_size columns (0: the range divided evenly over the handles), every slice an export
 *   of its own — blocks end at slice edges, as they do in hal2mafMP's output — dealt to the devices as they become free, the
 *   texts put together in input order with the header of the first slice only. */
int hgx_alignment_depth_multi(hgx_alignment *const *handles, int n_handles, int ref_genome, int ref_sequence, int64_t start,
                              int64_t length, int64_t step, int count_dupes, int no_ancestors, const int32_t *targets, int32_t n_targets,
                              char **out_text, size_t *out_len, char **err);
int hgx_maf_export_multi(hgx_alignment *const *handles, int n_handles, int ref_genome, int ref_sequence, int64_t start, int64_t length,
                         int64_t slice_size, const hgx_maf_opts *opts, const int32_t *targets, int32_t n_targets, char **out_text,
                         size_t *out_len, char **err);

/* hal2maf's plain export takes the columns where rows begin and end from per-base tracks made by sweeps over whole genomes
 * (hal_amd/csrc/hgx_maf_kernels.hpp) and keeps them with the handle for the export's chunks and for later exports of the same
 * reference, scope and filters.  *json (release with hgx_free): {"tracks": bool, "build_ms", "bytes", "state": "unchecked" |
 * "checked against the column walk" | "refused: ...", "chunks_served", ..., "last_export": the host side of this process's last
 * run-compressed export (who walked its blocks — one thread, or slices of the export side by side —, rounds, seconds) or null};
 * drop != 0 releases the tracks' device memory afterwards. */
int hgx_maf_tracks_info(hgx_alignment *h, int drop, char **json, char **err);

/* ---- synthetic workloads: halRandGen (randgen/halRandGen.cpp) ---- */
typedef struct hgx_rand_opts {
    double mean_degree, max_branch_length;
    uint64_t min_genomes, max_genomes, min_segment_length, max_segment_length, min_segments, max_segments;
    int32_t seed;
    int32_t with_dna; /* 1 = seed-compatible with halRandGen; 0 = skip DNA draws (faster, different stream); 2 = the
                         alignment of 0 with DNA from a separate fast generator (same model, seconds for 1 Gb) */
} hgx_rand_opts;
int hgx_rand_preset(const char *preset, hgx_rand_opts *opts); /* small | medium | big | large */
int hgx_create_random(const hgx_rand_opts *opts, int device, hgx_alignment **out, char **err);
int hgx_save_image(const hgx_alignment *h, const char *path, char **err);

void hgx_free(void *p);
/* The library keeps up to two released large texts (MAF, BED, wig: a gigabyte at most) and the page-locked host memory of its
 * copies for the next call; a long-lived embedder gives them back with this (nothing in use is touched).  A handle's own device
 * memory — its tables, and the per-base tracks hal2maf builds for a reference genome: five bytes a base of every genome in scope —
 * stays with the handle: hgx_maf_tracks_info(h, 1, ...) lets go of the tracks, hgx_close of everything. */
void hgx_release_cached(void);
/* "hgx <major>.<minor> (...)".  ABI history: 0.2 — hgx_maf_opts begins with struct_size (callers built against 0.1 must be
 * rebuilt: every field moved by four bytes, and an options struct without struct_size is refused), hgx_release_cached. */
const char *hgx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HGX_H */
