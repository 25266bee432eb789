"""Loads libhgx.so and declares the C ABI of include/hgx.h for ctypes."""
import ctypes as C
import os

# (HGX_LIB_PATH: an instrumented build, e.g. hal_amd/libhgx_prof.so of `make profile-lib`)
LIB_PATH = os.environ.get("HGX_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhgx.so")


class HgxError(RuntimeError):
    pass


class hgx_interval(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("seq", C.c_int32), ("strand", C.c_char), ("_pad", C.c_char * 3)]


class hgx_record(C.Structure):
    _fields_ = [("query", C.c_int64), ("tgt_start", C.c_int64), ("tgt_end", C.c_int64), ("src_start", C.c_int64),
                ("tgt_seq", C.c_int32), ("strand", C.c_char), ("tgt_reversed", C.c_uint8), ("_pad", C.c_char * 2)]


class hgx_liftover_opts(C.Structure):
    _fields_ = [("traverse_dupes", C.c_int32), ("coalescence_limit", C.c_int32), ("min_length", C.c_int64),
                ("emit_blocks", C.c_int32), ("block_mapper_source", C.c_int32)]


class hgx_liftover_stats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("source_pieces", C.c_uint64), ("top_derefs", C.c_uint64),
                ("bottom_derefs", C.c_uint64), ("mapped_pieces", C.c_uint64), ("records", C.c_uint64),
                ("deferred_queries", C.c_uint64), ("walk_ms", C.c_double), ("total_ms", C.c_double),
                ("composed_records", C.c_uint64), ("composed_build_ms", C.c_double),
                ("composed_kind", C.c_uint64), ("general_queries", C.c_uint64), ("composed_flagged", C.c_uint64)]


class hgx_column_opts(C.Structure):
    _fields_ = [("no_dupes", C.c_int32), ("no_ancestors", C.c_int32), ("only_orthologs", C.c_int32), ("n_targets", C.c_int32),
                ("targets", C.POINTER(C.c_int32))]


class hgx_column_row(C.Structure):
    _fields_ = [("pos", C.c_int64), ("genome", C.c_int32), ("reversed", C.c_uint8), ("base", C.c_char), ("_pad", C.c_uint8 * 2)]


class hgx_maf_opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("no_dupes", C.c_int32), ("no_ancestors", C.c_int32), ("only_sequence_names", C.c_int32),
                ("only_orthologs", C.c_int32), ("keep_empty_ref_blocks", C.c_int32), ("unique", C.c_int32),
                ("max_block_len", C.c_int64), ("max_ref_gap", C.c_int64), ("print_tree", C.c_int32)]


def maf_opts(**kw):
    """hgx_maf_opts with struct_size set (HGX_MAF_OPTS_INIT) and the given fields; booleans become 0 / 1"""
    o = hgx_maf_opts()
    o.struct_size = C.sizeof(hgx_maf_opts)
    for k, v in kw.items():
        setattr(o, k, int(v))
    return o


class hgx_rand_opts(C.Structure):
    _fields_ = [("mean_degree", C.c_double), ("max_branch_length", C.c_double), ("min_genomes", C.c_uint64),
                ("max_genomes", C.c_uint64), ("min_segment_length", C.c_uint64), ("max_segment_length", C.c_uint64),
                ("min_segments", C.c_uint64), ("max_segments", C.c_uint64), ("seed", C.c_int32), ("with_dna", C.c_int32)]


# name -> (restype, argtypes); every symbol include/hgx.h declares
P = C.POINTER
VP = C.c_void_p
ERR = P(C.c_char_p)
SYMBOLS = {
    "hgx_open": (C.c_int, [C.c_char_p, C.c_int, P(VP), P(VP)]),
    "hgx_close": (None, [VP]),
    "hgx_builder_begin": (C.c_int, [P(VP), P(VP)]),
    "hgx_builder_add_genome": (C.c_int, [VP, C.c_char_p, C.c_char_p, C.c_double, C.c_int64, P(C.c_char_p), P(C.c_int64),
                                         P(C.c_int64), P(C.c_int64), C.c_int64, P(C.c_int64), P(C.c_int64), P(C.c_uint8),
                                         P(C.c_int64), P(C.c_int64), C.c_int64, P(C.c_int64), P(C.c_int64), C.c_int64,
                                         P(C.c_int64), P(C.c_uint8), C.c_char_p, P(VP)]),
    "hgx_builder_finish": (C.c_int, [VP, C.c_int, P(VP), P(VP)]),
    "hgx_builder_abort": (None, [VP]),
    "hgx_num_genomes": (C.c_int, [VP]),
    "hgx_newick": (C.c_char_p, [VP]),
    "hgx_genome_name": (C.c_char_p, [VP, C.c_int]),
    "hgx_genome_id": (C.c_int, [VP, C.c_char_p]),
    "hgx_genome_parent": (C.c_int, [VP, C.c_int]),
    "hgx_genome_num_children": (C.c_int, [VP, C.c_int]),
    "hgx_genome_child": (C.c_int, [VP, C.c_int, C.c_int]),
    "hgx_genome_length": (C.c_int64, [VP, C.c_int]),
    "hgx_genome_num_top": (C.c_int64, [VP, C.c_int]),
    "hgx_genome_num_bottom": (C.c_int64, [VP, C.c_int]),
    "hgx_genome_num_sequences": (C.c_int, [VP, C.c_int]),
    "hgx_sequence_info": (C.c_int, [VP, C.c_int, C.c_int, P(C.c_char_p), P(C.c_int64), P(C.c_int64)]),
    "hgx_sequence_lookup": (C.c_int, [VP, C.c_int, C.c_char_p, P(C.c_int64), P(C.c_int64)]),
    "hgx_mrca": (C.c_int, [VP, C.c_int, C.c_int]),
    "hgx_liftover_batch": (C.c_int, [VP, C.c_int, C.c_int, C.c_size_t, P(hgx_interval), P(hgx_liftover_opts),
                                     P(P(hgx_record)), P(C.c_size_t), P(VP)]),
    "hgx_block_map": (C.c_int, [VP, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, P(P(hgx_record)),
                                P(C.c_size_t), P(VP)]),
    "hgx_maf_export_global": (C.c_int, [VP, P(hgx_maf_opts), P(VP), P(C.c_size_t), P(VP)]),
    "hgx_alignment_depth_multi": (C.c_int, [P(VP), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, P(C.c_int32), C.c_int32,
                                             P(VP), P(C.c_size_t), P(VP)]),
    "hgx_maf_export_multi": (C.c_int, [P(VP), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, P(hgx_maf_opts), P(C.c_int32), C.c_int32,
                                        P(VP), P(C.c_size_t), P(VP)]),
    "hgx_get_blocks_in_target_range": (VP, [VP, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                            C.c_char_p, P(VP)]),
    "hgx_get_blocks_in_target_ranges": (C.c_int, [VP, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, P(C.c_int64), P(C.c_int64), C.c_int64,
                                                  C.c_int, C.c_int, C.c_int, C.c_char_p, P(VP), P(VP)]),
    "hgx_free_block_results": (None, [VP]),
    "hgx_columns_depth_stats": (C.c_int, [VP, C.c_int, C.c_int64, C.c_int64, C.c_int64, P(hgx_column_opts), P(C.c_uint64), P(C.c_uint64),
                                          P(VP)]),
    "hgx_liftover_plan_set_timing": (C.c_int, [VP, C.c_int]),
    "hgx_liftover_plan_set_workers": (C.c_int, [VP, C.c_int]),
    "hgx_liftover_plan_create": (C.c_int, [VP, C.c_int, C.c_int, P(hgx_liftover_opts), C.c_size_t, P(VP), P(VP)]),
    "hgx_liftover_plan_destroy": (None, [VP]),
    "hgx_liftover_run_device": (C.c_int, [VP, C.c_size_t, VP, VP, VP, VP, P(VP), P(C.c_size_t), P(VP)]),
    "hgx_liftover_submit": (C.c_int, [VP, C.c_size_t, VP, VP, VP, VP, P(VP)]),
    "hgx_liftover_collect": (C.c_int, [VP, P(VP), P(C.c_size_t), P(VP)]),
    "hgx_liftover_last_stats": (C.c_int, [VP, P(hgx_liftover_stats)]),
    "hgx_clone_to_device": (C.c_int, [VP, C.c_int, P(VP), P(VP)]),
    "hgx_comm_unique_id": (C.c_int, [VP, P(VP)]),
    "hgx_comm_create": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, P(VP), P(VP)]),
    "hgx_comm_destroy": (None, [VP]),
    "hgx_liftover_exchange": (C.c_int, [VP, VP, C.c_int64, VP, C.c_size_t, VP, P(C.c_size_t), P(VP)]),
    "hgx_liftover_gather": (C.c_int, [VP, VP, C.c_int, C.c_int64, VP, C.c_size_t, C.c_int, VP, P(C.c_size_t), P(VP)]),
    "hgx_liftover_render_blobs": (C.c_int, [VP, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_int, P(VP), P(C.c_size_t), C.c_int, P(VP), P(C.c_size_t), P(VP)]),
    "hgx_liftover_gather_writers": (C.c_int, [VP, VP, C.c_int, C.c_int64, VP, C.c_size_t, C.c_int, VP, P(C.c_size_t), P(VP)]),
    "hgx_comm_all_sizes": (C.c_int, [VP, C.c_uint64, P(C.c_uint64), P(VP)]),
    "hgx_liftover_convert_multi": (C.c_int, [P(VP), C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             P(VP), P(C.c_size_t), P(VP)]),
    "hgx_liftover_kernel_times": (C.c_int, [VP, P(VP)]),
    "hgx_liftover_build_phases": (C.c_int, [P(VP)]),
    "hgx_liftover_copy_records": (C.c_int, [VP, VP, C.c_size_t, VP, P(VP)]),
    "hgx_liftover_copy_records_packed": (C.c_int, [VP, VP, C.c_size_t, VP, P(VP)]),
    "hgx_liftover_wire_blob": (C.c_int, [VP, VP, C.c_size_t, C.c_int64, P(C.c_size_t), P(C.c_int), VP, P(VP)]),
    "hgx_liftover_convert": (C.c_int, [VP, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, P(VP), P(C.c_size_t), P(VP)]),
    "hgx_columns_depth": (C.c_int, [VP, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, P(hgx_column_opts), P(C.c_int32), P(VP)]),
    "hgx_columns_depth_device": (C.c_int, [VP, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, P(hgx_column_opts), VP, VP,
                                           P(C.c_double), P(VP)]),
    "hgx_column_rows": (C.c_int, [VP, C.c_int, C.c_int64, C.c_int64, P(hgx_column_opts), P(P(C.c_uint64)), P(P(hgx_column_row)),
                                  P(C.c_size_t), P(VP)]),
    "hgx_alignment_depth": (C.c_int, [VP, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, P(C.c_int32), C.c_int32,
                                      P(VP), P(C.c_size_t), P(VP)]),
    "hgx_maf_export": (C.c_int, [VP, C.c_int, C.c_int, C.c_int64, C.c_int64, P(hgx_maf_opts), P(C.c_int32), C.c_int32, P(VP),
                                 P(C.c_size_t), P(VP)]),
    "hgx_maf_export_bed": (C.c_int, [VP, C.c_int, C.c_char_p, C.c_size_t, P(hgx_maf_opts), P(C.c_int32), C.c_int32, P(VP), P(C.c_size_t),
                                     P(VP)]),
    "hgx_rand_preset": (C.c_int, [C.c_char_p, P(hgx_rand_opts)]),
    "hgx_create_random": (C.c_int, [P(hgx_rand_opts), C.c_int, P(VP), P(VP)]),
    "hgx_save_image": (C.c_int, [VP, C.c_char_p, P(VP)]),
    "hgx_free": (None, [VP]),
    "hgx_release_cached": (None, []),
    "hgx_maf_tracks_info": (C.c_int, [VP, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "hgx_version": (C.c_char_p, []),
}


def _load():
    # PyTorch-ROCm bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  Two
    # runtimes in one process cannot both own the GPU, so when torch is installed it is imported first and
    # libhgx.so binds to the runtime torch loaded; device pointers and streams are then shared with torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise HgxError(
            "%s is missing: the HIP extension has not been built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hal_amd/csrc lib`).  hal_amd has no CPU fallback." % LIB_PATH)
    dll = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(dll, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return dll


lib = _load()


def take_error(err):
    """Consume a `char **err` out-parameter: return the message and free it."""
    if not err.value:
        return "unknown error"
    msg = C.string_at(err.value).decode(errors="replace")
    lib.hgx_free(err)
    return msg
