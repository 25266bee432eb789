"""Python mirror of the reference interfaces this library stands behind (argument names and meaning
follow Liftover::convert, liftover/inc/halLiftover.h:25-28, and the Alignment/Genome getters)."""
import time
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import (lib, HgxError, hgx_interval, hgx_record, hgx_liftover_opts, hgx_liftover_stats, hgx_rand_opts, hgx_column_opts,
                   hgx_column_row, hgx_maf_opts, maf_opts, take_error)

RECORD_DTYPE = np.dtype([("query", "<i8"), ("tgt_start", "<i8"), ("tgt_end", "<i8"), ("src_start", "<i8"),
                         ("tgt_seq", "<i4"), ("strand", "S1"), ("tgt_reversed", "u1"), ("_pad", "S2")])
assert RECORD_DTYPE.itemsize == C.sizeof(hgx_record) == 40
INTERVAL_DTYPE = np.dtype([("start", "<i8"), ("end", "<i8"), ("seq", "<i4"), ("strand", "S1"), ("_pad", "S3")])
assert INTERVAL_DTYPE.itemsize == C.sizeof(hgx_interval) == 24


@dataclass
class Interval:
    seq: int
    start: int
    end: int
    strand: str = "+"


@dataclass
class Record:
    query: int
    tgt_seq: int
    tgt_start: int
    tgt_end: int
    strand: str
    src_start: int


@dataclass
class RandOptions:
    """halRandGen options (randgen/halRandGen.cpp:39-56)."""
    mean_degree: float = 1.25
    max_branch_length: float = 0.7
    min_genomes: int = 8
    max_genomes: int = 20
    min_segment_length: int = 500
    max_segment_length: int = 2000
    min_segments: int = 100
    max_segments: int = 500
    seed: int = -1
    with_dna: object = True  # True: halRandGen's stream; False: no DNA (different, faster stream); "fast": False's alignment + DNA from a fast generator

    @staticmethod
    def preset(name, seed=-1, with_dna=True, **overrides):
        """halRandGen --preset <name> [--minSegmentLength ...]: the preset's values, then the options given (randgen/halRandGen.cpp:58-108)"""
        o = hgx_rand_opts()
        o.seed = seed
        o.with_dna = 1 if with_dna else 0
        if lib.hgx_rand_preset(name.encode(), C.byref(o)) != 0:
            raise HgxError("invalid --preset value: %s" % name)
        r = RandOptions(o.mean_degree, o.max_branch_length, o.min_genomes, o.max_genomes, o.min_segment_length,
                        o.max_segment_length, o.min_segments, o.max_segments, seed, with_dna)
        for k, v in overrides.items():
            if not hasattr(r, k):
                raise HgxError("no such halRandGen option: %s" % k)
            setattr(r, k, v)
        return r

    def _c(self):
        o = hgx_rand_opts()
        (o.mean_degree, o.max_branch_length, o.min_genomes, o.max_genomes, o.min_segment_length, o.max_segment_length,
         o.min_segments, o.max_segments, o.seed, o.with_dna) = (
            self.mean_degree, self.max_branch_length, self.min_genomes, self.max_genomes, self.min_segment_length,
            self.max_segment_length, self.min_segments, self.max_segments, self.seed,
            2 if self.with_dna == "fast" else (1 if self.with_dna else 0))
        return o


class Alignment:
    """An open alignment (openHalAlignment, api/impl/halAlignmentInstance.cpp:133-165).  device=-1 keeps the
    tables on the host (metadata only); device>=0 uploads them to that GPU."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def open(path, device=0):
        h, err = C.c_void_p(), C.c_void_p()
        if lib.hgx_open(str(path).encode(), device, C.byref(h), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return Alignment(h)

    @staticmethod
    def random(opts, device=0):
        h, err = C.c_void_p(), C.c_void_p()
        o = opts._c()
        if lib.hgx_create_random(C.byref(o), device, C.byref(h), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return Alignment(h)

    def close(self):
        if self._h:
            lib.hgx_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone_to_device(self, device):
        """Another handle of this alignment with its tables on `device` (hgx_clone_to_device): the host image is shared."""
        out, err = C.c_void_p(), C.c_void_p()
        if lib.hgx_clone_to_device(self._h, device, C.byref(out), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return Alignment(out)

    def save(self, path):
        err = C.c_void_p()
        if lib.hgx_save_image(self._h, str(path).encode(), C.byref(err)) != 0:
            raise HgxError(take_error(err))

    # --- Alignment / Genome / Sequence getters ---
    @property
    def num_genomes(self):
        return lib.hgx_num_genomes(self._h)

    @property
    def newick(self):
        return lib.hgx_newick(self._h).decode()

    def genome_name(self, g):
        n = lib.hgx_genome_name(self._h, g)
        if n is None:
            raise HgxError("genome id out of range")
        return n.decode()

    def genome_id(self, name):
        return lib.hgx_genome_id(self._h, name.encode())

    def genome_parent(self, g):
        return lib.hgx_genome_parent(self._h, g)

    def genome_children(self, g):
        return [lib.hgx_genome_child(self._h, g, k) for k in range(lib.hgx_genome_num_children(self._h, g))]

    def genome_length(self, g):
        return lib.hgx_genome_length(self._h, g)

    def num_top_segments(self, g):
        return lib.hgx_genome_num_top(self._h, g)

    def num_bottom_segments(self, g):
        return lib.hgx_genome_num_bottom(self._h, g)

    def sequences(self, g):
        out = []
        for s in range(lib.hgx_genome_num_sequences(self._h, g)):
            name, start, length = C.c_char_p(), C.c_int64(), C.c_int64()
            lib.hgx_sequence_info(self._h, g, s, C.byref(name), C.byref(start), C.byref(length))
            out.append((name.value.decode(), start.value, length.value))
        return out

    def sequence_lookup(self, g, name):
        start, length = C.c_int64(), C.c_int64()
        s = lib.hgx_sequence_lookup(self._h, g, name.encode(), C.byref(start), C.byref(length))
        return (s, start.value, length.value) if s >= 0 else None

    def mrca(self, a, b):
        return lib.hgx_mrca(self._h, a, b)

    # --- liftover, host-buffer form ---
    def columns_depth_stats(self, ref, first, count, step=1, **col_opts):
        """{top_derefs, bottom_derefs}: segment records the column walks of columns_depth(...) logically dereference."""
        o, keep = self._column_opts(**col_opts)
        t, b, err = C.c_uint64(), C.c_uint64(), C.c_void_p()
        if lib.hgx_columns_depth_stats(self._h, ref, first, count, step, C.byref(o), C.byref(t), C.byref(b), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return {"top_derefs": t.value, "bottom_derefs": b.value}

    def block_map(self, ref, query, abs_first, abs_last, target_reversed=False, do_dupes=True, min_length=0, coalescence_limit=-1):
        """BlockMapper init + map + getMap without adjacencies (liftover/inc/halBlockMapper.h:30-40): the members of the mapped
        set for the reference range [abs_first, abs_last] (genome coordinates, inclusive), in set order, as a numpy record
        array (RECORD_DTYPE; see hgx_liftover_opts.emit_blocks in include/hgx.h for the field meaning)."""
        out, n, err = C.POINTER(hgx_record)(), C.c_size_t(), C.c_void_p()
        if lib.hgx_block_map(self._h, ref, query, abs_first, abs_last, 1 if target_reversed else 0, 1 if do_dupes else 0, min_length,
                             coalescence_limit, C.byref(out), C.byref(n), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        try:
            return np.frombuffer(C.string_at(out, n.value * C.sizeof(hgx_record)), dtype=RECORD_DTYPE).copy()
        finally:
            lib.hgx_free(out)

    def blocks_in_target_ranges(self, q_species, t_species, t_chrom, ranges, t_reversed=False, seq=False, dup_mode=2, adjacencies=True,
                                coalescence_limit=None, decode=True):
        """halGetBlocksInTargetRange (blockViz/inc/halBlockViz.h:222-225) for every (t_start, t_end) of `ranges` in one call
        (hgx_get_blocks_in_target_ranges).  Per range: (blocks, target_dupes) — blocks = list of dicts with the fields of
        hal_block_t (qChrom, tStart, qStart, size, strand, qSequence, tSequence), target_dupes = list of (id, qChrom, [(tStart,
        size), ...]).  decode=False: the results are released unread and None is returned (benchmark use: the library's time)."""
        n = len(ranges)
        starts = (C.c_int64 * max(n, 1))(*[r[0] for r in ranges])
        ends = (C.c_int64 * max(n, 1))(*[r[1] for r in ranges])
        res = (C.c_void_p * max(n, 1))()
        err = C.c_void_p()
        lim = coalescence_limit.encode() if coalescence_limit is not None else None
        if lib.hgx_get_blocks_in_target_ranges(self._h, q_species.encode(), t_species.encode(), t_chrom.encode(), n, starts, ends,
                                               1 if t_reversed else 0, 1 if seq else 0, dup_mode, 1 if adjacencies else 0, lim,
                                               C.cast(res, C.POINTER(C.c_void_p)), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        out = []
        for k in range(n):
            try:
                if decode:
                    out.append(_read_block_results(res[k]))
            finally:
                lib.hgx_free_block_results(res[k])
        return out if decode else None

    def blocks_in_target_range(self, q_species, t_species, t_chrom, t_start, t_end, **kw):
        """halGetBlocksInTargetRange for one range: (blocks, target_dupes)"""
        return self.blocks_in_target_ranges(q_species, t_species, t_chrom, [(t_start, t_end)], **kw)[0]

    def liftover_batch(self, src, tgt, intervals, traverse_dupes=True, min_length=0, coalescence_limit=-1):
        """intervals: numpy array of INTERVAL_DTYPE or list of Interval.  Returns a numpy array of RECORD_DTYPE."""
        if not isinstance(intervals, np.ndarray):
            arr = np.zeros(len(intervals), dtype=INTERVAL_DTYPE)
            for i, q in enumerate(intervals):
                arr[i] = (q.start, q.end, q.seq, q.strand.encode(), b"")
            intervals = arr
        intervals = np.ascontiguousarray(intervals, dtype=INTERVAL_DTYPE)
        opts = hgx_liftover_opts(1 if traverse_dupes else 0, coalescence_limit, min_length)
        out, n, err = C.POINTER(hgx_record)(), C.c_size_t(), C.c_void_p()
        rc = lib.hgx_liftover_batch(self._h, src, tgt, len(intervals),
                                    intervals.ctypes.data_as(C.POINTER(hgx_interval)), C.byref(opts), C.byref(out),
                                    C.byref(n), C.byref(err))
        if rc != 0:
            raise HgxError(take_error(err))
        try:
            # one copy out of the library's buffer (string_at + frombuffer().copy() would be two)
            res = (np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n.value * 40,)).view(RECORD_DTYPE).copy()
                   if n.value else np.zeros(0, RECORD_DTYPE))
        finally:
            lib.hgx_free(out)
        return res


    # --- column engine (ColumnIterator defaults: halAlignmentDepth, hal2maf) ---
    @staticmethod
    def _column_opts(no_dupes=False, no_ancestors=False, only_orthologs=False, targets=None):
        o = hgx_column_opts(1 if no_dupes else 0, 1 if no_ancestors else 0, 1 if only_orthologs else 0, 0, None)
        keep = None
        if targets:
            keep = (C.c_int32 * len(targets))(*targets)
            o.n_targets = len(targets)
            o.targets = C.cast(keep, C.POINTER(C.c_int32))
        return o, keep

    def columns_depth(self, ref, first, count, step=1, count_dupes=False, **kw):
        """Per-column halAlignmentDepth value for columns first, first+step, ... (genome coordinates)."""
        o, keep = self._column_opts(**kw)
        out = np.empty(count, dtype=np.int32)
        err = C.c_void_p()
        if lib.hgx_columns_depth(self._h, ref, first, count, step, 1 if count_dupes else 0, C.byref(o),
                                 out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return out

    def columns_depth_device(self, ref, first, count, d_out_ptr, step=1, count_dupes=False, stream=0, **kw):
        """Result left in HBM at d_out_ptr (int32[count]); returns the kernel's device time in ms."""
        o, keep = self._column_opts(**kw)
        ms, err = C.c_double(), C.c_void_p()
        if lib.hgx_columns_depth_device(self._h, ref, first, count, step, 1 if count_dupes else 0, C.byref(o), d_out_ptr, stream,
                                        C.byref(ms), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return ms.value

    def column_rows(self, ref, first, count, **kw):
        """(row_offset[count+1], rows) in ColumnMap insertion order."""
        o, keep = self._column_opts(**kw)
        off, rows, n, err = C.POINTER(C.c_uint64)(), C.POINTER(hgx_column_row)(), C.c_size_t(), C.c_void_p()
        if lib.hgx_column_rows(self._h, ref, first, count, C.byref(o), C.byref(off), C.byref(rows), C.byref(n), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        try:
            offsets = np.frombuffer(C.string_at(off, (count + 1) * 8), dtype="<u8").copy()
            dt = np.dtype([("pos", "<i8"), ("genome", "<i4"), ("reversed", "u1"), ("base", "S1"), ("_pad", "S2")])
            r = np.frombuffer(C.string_at(rows, n.value * 16), dtype=dt).copy() if n.value else np.zeros(0, dt)
        finally:
            lib.hgx_free(off)
            lib.hgx_free(rows)
        return offsets, r

    def alignment_depth(self, ref, ref_sequence=-1, start=0, length=0, step=1, count_dupes=False, no_ancestors=False, targets=None):
        """halAlignmentDepth's wig text (alignmentDepth/halAlignmentDepth.cpp:318-347)."""
        tg = (C.c_int32 * len(targets))(*targets) if targets else None
        out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
        if lib.hgx_alignment_depth(self._h, ref, ref_sequence, start, length, step, 1 if count_dupes else 0, 1 if no_ancestors else 0,
                                   tg, len(targets) if targets else 0, C.byref(out), C.byref(n), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        try:
            return C.string_at(out, n.value).decode()
        finally:
            lib.hgx_free(out)

    def alignment_depth_bytes(self, ref, ref_sequence=-1, start=0, length=0, step=1, count_dupes=False, no_ancestors=False, prefix=0):
        """halAlignmentDepth end to end, the wig text left in library memory and released: returns its size (benchmark use);
        prefix > 0: (size, the text's first `prefix` bytes, seconds of the call itself) — what a parity check of a timed run looks at."""
        out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
        t0 = time.perf_counter()
        if lib.hgx_alignment_depth(self._h, ref, ref_sequence, start, length, step, 1 if count_dupes else 0, 1 if no_ancestors else 0,
                                   None, 0, C.byref(out), C.byref(n), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        seconds = time.perf_counter() - t0
        head = C.string_at(out, min(prefix, n.value)) if prefix > 0 and out.value else b""
        lib.hgx_free(out)
        return (n.value, head, seconds) if prefix > 0 else n.value

    def maf_export_bytes(self, ref, ref_sequence=-1, start=0, length=0, no_ancestors=False, max_block_len=1000, unique=False, max_ref_gap=0,
                         prefix=0):
        """hal2maf end to end, the text left in library memory and released: returns its size (benchmark use); prefix > 0: (size,
        the text's first `prefix` bytes, seconds of the call itself)."""
        o = maf_opts(no_ancestors=no_ancestors, max_block_len=max_block_len, unique=unique, max_ref_gap=max_ref_gap)
        out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
        t0 = time.perf_counter()
        if lib.hgx_maf_export(self._h, ref, ref_sequence, start, length, C.byref(o), None, 0, C.byref(out), C.byref(n), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        seconds = time.perf_counter() - t0
        head = C.string_at(out, min(prefix, n.value)) if prefix > 0 and out.value else b""
        lib.hgx_free(out)
        return (n.value, head, seconds) if prefix > 0 else n.value

    def maf_tracks_info(self, drop=False):
        """hal2maf's per-base tracks kept with this handle (hgx_maf_tracks_info): a dict; drop=True releases them afterwards"""
        import json
        out, err = C.c_void_p(), C.c_void_p()
        if lib.hgx_maf_tracks_info(self._h, 1 if drop else 0, C.byref(out), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        try:
            return json.loads(C.string_at(out).decode())
        finally:
            lib.hgx_free(out)

    def maf_export_global(self, no_dupes=False, no_ancestors=False, only_sequence_names=False, only_orthologs=False, max_block_len=1000,
                          print_tree=False):
        """hal2maf --global (MafExport::convertEntireAlignment, maf/impl/halMafExport.cpp:90-153): every column of the alignment once"""
        o = maf_opts(no_dupes=no_dupes, no_ancestors=no_ancestors, only_sequence_names=only_sequence_names, only_orthologs=only_orthologs,
                     max_block_len=max_block_len, print_tree=print_tree)
        out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
        if lib.hgx_maf_export_global(self._h, C.byref(o), C.byref(out), C.byref(n), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        try:
            return C.string_at(out, n.value).decode()
        finally:
            lib.hgx_free(out)

    def maf_export(self, ref, ref_sequence=-1, start=0, length=0, no_dupes=False, no_ancestors=False, only_sequence_names=False,
                   only_orthologs=False, keep_empty_ref_blocks=False, max_block_len=1000, targets=None, unique=False,
                   ref_targets_bed=None, max_ref_gap=0, print_tree=False):
        """hal2maf's MAF text (maf/impl/halMafExport.cpp:25-88, maf/impl/hal2maf.cpp:196-206); ref_targets_bed: BED text of
        reference intervals (--refTargets, maf/impl/halMafBed.cpp)."""
        if ref_targets_bed is not None:
            o = maf_opts(no_dupes=no_dupes, no_ancestors=no_ancestors, only_sequence_names=only_sequence_names, only_orthologs=only_orthologs,
                         keep_empty_ref_blocks=keep_empty_ref_blocks, unique=unique, max_block_len=max_block_len, max_ref_gap=max_ref_gap,
                         print_tree=print_tree)
            tg = (C.c_int32 * len(targets))(*targets) if targets else None
            data = ref_targets_bed.encode() if isinstance(ref_targets_bed, str) else ref_targets_bed
            out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
            if lib.hgx_maf_export_bed(self._h, ref, data, len(data), C.byref(o), tg, len(targets) if targets else 0, C.byref(out),
                                      C.byref(n), C.byref(err)) != 0:
                raise HgxError(take_error(err))
            try:
                return C.string_at(out, n.value).decode()
            finally:
                lib.hgx_free(out)
        o = maf_opts(no_dupes=no_dupes, no_ancestors=no_ancestors, only_sequence_names=only_sequence_names, only_orthologs=only_orthologs,
                         keep_empty_ref_blocks=keep_empty_ref_blocks, unique=unique, max_block_len=max_block_len, max_ref_gap=max_ref_gap,
                         print_tree=print_tree)
        tg = (C.c_int32 * len(targets))(*targets) if targets else None
        out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
        if lib.hgx_maf_export(self._h, ref, ref_sequence, start, length, C.byref(o), tg, len(targets) if targets else 0,
                              C.byref(out), C.byref(n), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        try:
            return C.string_at(out, n.value).decode()
        finally:
            lib.hgx_free(out)


def liftover_convert(alignment, src_genome, bed_text, tgt_genome, bed_type=0, traverse_dupes=True, out_psl=False,
                     out_psl_with_name=False, coalescence_limit=-1):
    """Liftover::convert (liftover/impl/halLiftover.cpp:23-41) on BED text; returns the output BED text.
    Raises HgxError with the reference's message on malformed input (after lifting the preceding lines)."""
    data = bed_text.encode() if isinstance(bed_text, str) else bed_text
    out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    rc = lib.hgx_liftover_convert(alignment._h, src_genome, data, len(data), tgt_genome, bed_type, 1 if traverse_dupes else 0,
                                  1 if out_psl else 0, 1 if out_psl_with_name else 0, coalescence_limit, C.byref(out),
                                  C.byref(n), C.byref(err))
    text = C.string_at(out, n.value).decode() if out.value else ""
    if out.value:
        lib.hgx_free(out)
    if rc != 0:
        e = HgxError(take_error(err))
        e.partial_output = text
        raise e
    return text


def liftover_render_blobs(alignment, src_genome, tgt_genome, bed_text, blobs, bed_type=0):
    """hgx_liftover_render_blobs: the lifted BED text of the intervals a writer's wire blobs hold (blobs: bytes-like objects or uint8
    tensors on the host, in the order of the group's ranks; bed_text: the group's input lines, in order).  Needs no device."""
    data = bed_text.encode() if isinstance(bed_text, str) else bytes(bed_text)
    raw = [bytes(b.cpu().numpy().tobytes()) if hasattr(b, "cpu") else bytes(b) for b in blobs]
    keep = [C.create_string_buffer(r, len(r)) for r in raw]
    ptrs = (C.c_void_p * len(keep))(*[C.cast(k, C.c_void_p) for k in keep])
    sizes = (C.c_size_t * len(keep))(*[len(r) for r in raw])
    out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    rc = lib.hgx_liftover_render_blobs(alignment._h, src_genome, tgt_genome, data, len(data), bed_type, ptrs, sizes, len(keep), C.byref(out),
                                       C.byref(n), C.byref(err))
    if rc != 0:
        raise HgxError(take_error(err))
    try:
        return C.string_at(out, n.value)
    finally:
        if out.value:
            lib.hgx_free(out)


def liftover_convert_multi(alignments, src_genome, bed_text, tgt_genome, bed_type=0, traverse_dupes=True, out_psl=False,
                           out_psl_with_name=False, coalescence_limit=-1):
    """hgx_liftover_convert_multi: Liftover::convert over several device clones of one alignment (Alignment.clone_to_device),
    the input's lines shared out over them; same result as liftover_convert."""
    data = bed_text.encode() if isinstance(bed_text, str) else bed_text
    hs = (C.c_void_p * len(alignments))(*[a._h for a in alignments])
    out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    rc = lib.hgx_liftover_convert_multi(hs, len(alignments), src_genome, data, len(data), tgt_genome, bed_type, 1 if traverse_dupes else 0,
                                        1 if out_psl else 0, 1 if out_psl_with_name else 0, coalescence_limit, C.byref(out), C.byref(n),
                                        C.byref(err))
    text = C.string_at(out, n.value).decode() if out.value else ""
    if out.value:
        lib.hgx_free(out)
    if rc != 0:
        e = HgxError(take_error(err))
        e.partial_output = text
        raise e
    return text


def alignment_depth_multi(alignments, ref, ref_sequence=-1, start=0, length=0, step=1, count_dupes=False, no_ancestors=False, targets=None):
    """hgx_alignment_depth_multi: halAlignmentDepth's wig text, the columns scanned in contiguous shares on several device clones of
    one alignment (Alignment.clone_to_device); same text as Alignment.alignment_depth."""
    hs = (C.c_void_p * len(alignments))(*[a._h for a in alignments])
    tg = (C.c_int32 * len(targets))(*targets) if targets else None
    out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    if lib.hgx_alignment_depth_multi(hs, len(alignments), ref, ref_sequence, start, length, step, 1 if count_dupes else 0,
                                     1 if no_ancestors else 0, tg, len(targets) if targets else 0, C.byref(out), C.byref(n), C.byref(err)) != 0:
        raise HgxError(take_error(err))
    try:
        return C.string_at(out, n.value).decode()
    finally:
        lib.hgx_free(out)


def maf_export_multi(alignments, ref, ref_sequence=-1, start=0, length=0, slice_size=0, no_dupes=False, no_ancestors=False,
                     only_sequence_names=False, only_orthologs=False, keep_empty_ref_blocks=False, max_block_len=1000, targets=None,
                     unique=False, max_ref_gap=0, print_tree=False, size_only=False):
    """hgx_maf_export_multi: hal2mafMP.py's slices (maf/hal2mafMP.py:63-79) — one export per slice of slice_size reference columns
    (0: the range divided evenly over the handles), dealt to the device clones, the texts concatenated with the first header only.
    size_only: the text is released unread and its size returned (benchmark use)."""
    hs = (C.c_void_p * len(alignments))(*[a._h for a in alignments])
    o = maf_opts(no_dupes=no_dupes, no_ancestors=no_ancestors, only_sequence_names=only_sequence_names, only_orthologs=only_orthologs,
                         keep_empty_ref_blocks=keep_empty_ref_blocks, unique=unique, max_block_len=max_block_len, max_ref_gap=max_ref_gap,
                         print_tree=print_tree)
    tg = (C.c_int32 * len(targets))(*targets) if targets else None
    out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    if lib.hgx_maf_export_multi(hs, len(alignments), ref, ref_sequence, start, length, slice_size, C.byref(o), tg,
                                len(targets) if targets else 0, C.byref(out), C.byref(n), C.byref(err)) != 0:
        raise HgxError(take_error(err))
    try:
        return n.value if size_only else C.string_at(out, n.value).decode()
    finally:
        lib.hgx_free(out)


def liftover_convert_bytes(alignment, src_genome, data, tgt_genome, bed_type=0, traverse_dupes=True, count_lines=True, out_psl=False):
    """hgx_liftover_convert on BED bytes, the output left in library memory and released: (bytes, lines) of it (benchmark use:
    no decoding of a hundred megabytes of text in Python)."""
    out, n, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
    rc = lib.hgx_liftover_convert(alignment._h, src_genome, data, len(data), tgt_genome, bed_type, 1 if traverse_dupes else 0,
                                  1 if out_psl else 0, 0, -1, C.byref(out), C.byref(n), C.byref(err))
    lines = C.string_at(out, n.value).count(b"\n") if (out.value and count_lines) else 0
    if out.value:
        lib.hgx_free(out)
    if rc != 0:
        raise HgxError(take_error(err))
    return n.value, lines


class Comm:
    """An RCCL communicator made by the library itself (hgx_comm_create): one process per GPU.  The 128-byte id comes from
    Comm.unique_id() on one rank and reaches the others by whatever the launcher offers (torch.distributed, MPI, a file)."""

    def __init__(self, unique_id, rank, n_ranks, device):
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        c, err = C.c_void_p(), C.c_void_p()
        if lib.hgx_comm_create(buf, rank, n_ranks, device, C.byref(c), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        self._c, self.rank, self.n_ranks = c, rank, n_ranks

    @staticmethod
    def unique_id():
        buf, err = (C.c_ubyte * 128)(), C.c_void_p()
        if lib.hgx_comm_unique_id(buf, C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return bytes(buf)

    def all_sizes(self, mine):
        """hgx_comm_all_sizes: every rank's value in rank order (the writers' text sizes: where each one's text begins)"""
        out, err = (C.c_uint64 * self.n_ranks)(), C.c_void_p()
        if lib.hgx_comm_all_sizes(self._c, int(mine), out, C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return list(out)

    def close(self):
        if self._c:
            lib.hgx_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        self.close()


class _TargetRange(C.Structure):
    pass


_TargetRange._fields_ = [("next", C.POINTER(_TargetRange)), ("tStart", C.c_int64), ("size", C.c_int64)]


class _TargetDupe(C.Structure):
    pass


_TargetDupe._fields_ = [("next", C.POINTER(_TargetDupe)), ("id", C.c_int64), ("tRange", C.POINTER(_TargetRange)), ("qChrom", C.c_char_p)]


class _Block(C.Structure):
    pass


_Block._fields_ = [("next", C.POINTER(_Block)), ("qChrom", C.c_char_p), ("tStart", C.c_int64), ("qStart", C.c_int64), ("size", C.c_int64),
                   ("strand", C.c_char), ("qSequence", C.c_char_p), ("tSequence", C.c_char_p)]


class _BlockResults(C.Structure):
    _fields_ = [("mappedBlocks", C.POINTER(_Block)), ("targetDupeBlocks", C.POINTER(_TargetDupe))]


def _read_block_results(ptr):
    r = C.cast(ptr, C.POINTER(_BlockResults)).contents
    blocks, dupes = [], []
    b = r.mappedBlocks
    while b:
        x = b.contents
        blocks.append({"qChrom": x.qChrom.decode(), "tStart": x.tStart, "qStart": x.qStart, "size": x.size, "strand": x.strand.decode(),
                       "qSequence": x.qSequence.decode() if x.qSequence is not None else None,
                       "tSequence": x.tSequence.decode() if x.tSequence is not None else None})
        b = x.next
    d = r.targetDupeBlocks
    while d:
        x = d.contents
        ranges, t = [], x.tRange
        while t:
            ranges.append((t.contents.tStart, t.contents.size))
            t = t.contents.next
        dupes.append((x.id, x.qChrom.decode(), ranges))
        d = x.next
    return blocks, dupes


def format_block_results(blocks, dupes):
    """what blockVizTest --verbose prints for a result (blockViz/tests/blockVizTest.cpp:103-113)"""
    out = []
    for b in blocks:
        out.append("chr:%s, tSt:%d, qSt:%d, size:%d, strand:%s: tgt : %s query: %s\n"
                   % (b["qChrom"], b["tStart"], b["qStart"], b["size"], b["strand"],
                      "(null)" if b["tSequence"] is None else b["tSequence"][:10], "(null)" if b["qSequence"] is None else b["qSequence"][:10]))
    for did, chrom, ranges in dupes:
        out.append("tDupe id:%d qCrhom:%s\n" % (did, chrom))
        out += [" tSt:%d size:%d\n" % r for r in ranges]
    return "".join(out)


def build_phases():
    """[(phase, ms), ...] of this process's last table build made with HGX_BUILD_TIMING set (hgx_liftover_build_phases)"""
    import json
    js = C.c_void_p()
    if lib.hgx_liftover_build_phases(C.byref(js)) != 0:
        return []
    try:
        return [tuple(x) for x in json.loads(C.string_at(js.value).decode())]
    finally:
        lib.hgx_free(js)


class LiftoverPlan:
    """Device-resident liftover: queries and records stay in HBM (hgx_liftover_run_device)."""

    def __init__(self, alignment, src, tgt, max_queries, traverse_dupes=True, min_length=0):
        self._al = alignment
        opts = hgx_liftover_opts(1 if traverse_dupes else 0, -1, min_length)
        p, err = C.c_void_p(), C.c_void_p()
        if lib.hgx_liftover_plan_create(alignment._h, src, tgt, C.byref(opts), max_queries, C.byref(p), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        self._p = p

    def close(self):
        if self._p:
            lib.hgx_liftover_plan_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run_ptr(self, n, d_gstart, d_gend, d_strand, stream=0):
        """Raw-pointer form: device pointers (ints) of int64 start/end (inclusive genome coordinates) and uint8
        strand.  Returns (device pointer of hgx_record[n_records], n_records)."""
        out, nrec, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
        rc = lib.hgx_liftover_run_device(self._p, n, d_gstart, d_gend, d_strand, stream, C.byref(out), C.byref(nrec), C.byref(err))
        if rc != 0:
            raise HgxError(take_error(err))
        return out.value or 0, nrec.value

    def run(self, gstart, gend, strand):
        """torch tensors on the alignment's device: int64 gstart/gend, uint8 strand; runs on torch's current stream."""
        import torch
        assert gstart.is_cuda and gstart.dtype == torch.int64 and gend.dtype == torch.int64 and strand.dtype == torch.uint8
        stream = torch.cuda.current_stream(gstart.device).cuda_stream
        return self.run_ptr(gstart.numel(), gstart.data_ptr(), gend.data_ptr(), strand.data_ptr(), stream)

    def submit(self, gstart, gend, strand, stream=None):
        """hgx_liftover_submit: queue the batch on `stream` (a torch.cuda.Stream; default: the current one) and return; collect()
        waits for it.  The tensors must stay alive and unchanged until then."""
        import torch
        assert gstart.is_cuda and gstart.dtype == torch.int64 and gend.dtype == torch.int64 and strand.dtype == torch.uint8
        st = (stream or torch.cuda.current_stream(gstart.device)).cuda_stream
        err = C.c_void_p()
        if lib.hgx_liftover_submit(self._p, gstart.numel(), gstart.data_ptr(), gend.data_ptr(), strand.data_ptr(), st, C.byref(err)) != 0:
            raise HgxError(take_error(err))
        self._pending = (gstart, gend, strand)

    def collect(self):
        """hgx_liftover_collect: (device pointer of hgx_record[n_records], n_records) of the submitted batch"""
        out, nrec, err = C.c_void_p(), C.c_size_t(), C.c_void_p()
        if lib.hgx_liftover_collect(self._p, C.byref(out), C.byref(nrec), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        self._pending = None
        return out.value or 0, nrec.value

    def records_to_tensor(self, ptr, n, packed=False):
        """Copy the plan-owned device records of the last run into a fresh torch uint8 tensor (device to device): [n, 40]
        hgx_record rows, or with packed=True [n, 20] rows in the wire form of hal_amd.shard.pack_records."""
        import torch
        t = torch.empty((n, 20 if packed else 40), dtype=torch.uint8, device="cuda")
        err = C.c_void_p()
        stream = torch.cuda.current_stream().cuda_stream
        f = lib.hgx_liftover_copy_records_packed if packed else lib.hgx_liftover_copy_records
        if f(self._p, t.data_ptr(), n, stream, C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return t

    def wire_capacity(self):
        """bytes a destination of wire_blob needs for the last run's records (any of the three formats)"""
        st = self.stats()
        return 32 + (2 * st["queries"] + 7) // 8 * 8 + 40 * st["records"]

    def exchange(self, comm, first_query, gathered, slot_bytes):
        """hgx_liftover_exchange: this rank's records of the last run into slot comm.rank of `gathered` (uint8 device tensor of
        comm.n_ranks * slot_bytes) and one RCCL all-gather of the slots, ordered on the current stream.  Returns this rank's bytes."""
        import torch
        err, nbytes = C.c_void_p(), C.c_size_t()
        if lib.hgx_liftover_exchange(self._p, comm._c, first_query, gathered.data_ptr(), slot_bytes, torch.cuda.current_stream().cuda_stream,
                                     C.byref(nbytes), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return nbytes.value

    def gather(self, comm, root, first_query, gathered, slot_bytes, bed_only=False):
        """hgx_liftover_gather: this rank's records of the last run to rank `root` only (RCCL send / recv from the library, ordered
        on the current stream).  gathered: comm.n_ranks * slot_bytes on the root, one slot on the others.  Returns this rank's bytes."""
        import torch
        err, nbytes = C.c_void_p(), C.c_size_t()
        if lib.hgx_liftover_gather(self._p, comm._c, root, first_query, gathered.data_ptr(), slot_bytes, 1 if bed_only else 0,
                                   torch.cuda.current_stream().cuda_stream, C.byref(nbytes), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return nbytes.value

    def gather_writers(self, comm, group_size, first_query, gathered, slot_bytes, bed_only=False):
        """hgx_liftover_gather_writers: this rank's records of the last run to its group's writer (the first of group_size
        consecutive ranks).  gathered: group_size * slot_bytes on a writer, one slot on the others.  Returns this rank's bytes."""
        import torch
        err, nbytes = C.c_void_p(), C.c_size_t()
        if lib.hgx_liftover_gather_writers(self._p, comm._c, group_size, first_query, gathered.data_ptr(), slot_bytes, 1 if bed_only else 0,
                                           torch.cuda.current_stream().cuda_stream, C.byref(nbytes), C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return nbytes.value

    def wire_blob(self, first_query=0, dst=None, bed_only=False):
        """The last run's records as one self-describing uint8 tensor for the multi-GPU exchange (hgx_liftover_wire_blob;
        hal_amd.shard.decode_blob reads it): (blob, format) with format 12 or 20 bytes per record.  dst: write into this uint8
        device tensor (a slot of an exchange buffer) instead of a fresh one.  bed_only: the 8-byte form (no source coordinates: all a
        writer of BED lines needs) when the batch fits it."""
        import torch
        err, nbytes, fmt = C.c_void_p(), C.c_size_t(), C.c_int(8 if bed_only else 0)
        stream = torch.cuda.current_stream().cuda_stream
        t = dst if dst is not None else torch.empty(self.wire_capacity(), dtype=torch.uint8, device="cuda")
        if lib.hgx_liftover_wire_blob(self._p, t.data_ptr(), t.numel(), first_query, C.byref(nbytes), C.byref(fmt), stream, C.byref(err)) != 0:
            raise HgxError(take_error(err))
        return t[:nbytes.value], fmt.value

    def stats(self):
        s = hgx_liftover_stats()
        lib.hgx_liftover_last_stats(self._p, C.byref(s))
        return {k: getattr(s, k) for k, _ in hgx_liftover_stats._fields_}

    def set_timing(self, mode):
        """0: no per-kernel events; 1: kernel_times() covers the last run (default); 2: every run since the previous read."""
        if lib.hgx_liftover_plan_set_timing(self._p, mode) != 0:
            raise HgxError("set_timing failed")

    def set_workers(self, n):
        """hgx_liftover_plan_set_workers: < 0 the general intervals of a batch that runs by itself are found up front and finished by
        workgroups at the head of the classifying launch when the last batch had few (default); 0 never; n > 0 always, n workgroups."""
        if lib.hgx_liftover_plan_set_workers(self._p, n) != 0:
            raise HgxError("set_workers failed")

    def kernel_times(self):
        import json
        js = C.c_void_p()
        if lib.hgx_liftover_kernel_times(self._p, C.byref(js)) != 0:
            raise HgxError("kernel_times failed (the library says why on stderr)")
        try:
            return json.loads(C.string_at(js.value).decode())
        finally:
            lib.hgx_free(js)
