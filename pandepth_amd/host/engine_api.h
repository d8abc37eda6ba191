// engine_api.h — the depth-engine seam as a table of C function pointers with exactly the
// signatures of include/pandepth_amd.h.  The `pandepth` binary fills it with the gfx950 library's
// entry points (main.cpp) and nothing else; the CPU test harness under tests/ fills it with an
// oracle-backed implementation so that the host logic (readers, read selection, region model,
// table writer) can be checked byte-for-byte against the reference's golden files without a GPU.
#ifndef PD_ENGINE_API_H_
#define PD_ENGINE_API_H_
#include "../../include/pandepth_amd.h"
#include "../../include/pandepth_amd_dev.h"   /* pd_bgzf_unit of the optional one-call decode entry (tests) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pd_engine_api {
    int (*create)(int device, int32_t n_contigs, const uint32_t *contig_len, pd_ctx **out);
    int (*destroy)(pd_ctx *);
    const char *(*strerror)(const pd_ctx *);
    int (*push_intervals)(pd_ctx *, const pd_iv *, size_t, unsigned flags);
    int (*scan)(pd_ctx *, unsigned wrap_bits);
    int (*reduce_intervals)(pd_ctx *, const pd_region *, size_t, uint32_t, int32_t *, uint64_t *);
    int (*window_layout)(const pd_ctx *, uint32_t, uint64_t *);
    int (*scan_reduce_windows)(pd_ctx *, uint32_t, uint32_t, unsigned, uint32_t *, uint64_t *);
    int (*reduce_windows)(pd_ctx *, uint32_t, uint32_t, uint32_t *, uint64_t *);
    int (*read_depth)(pd_ctx *, int32_t, uint32_t, size_t, uint32_t *);
    int (*synchronize)(pd_ctx *);
    /* optional (NULL = not available): GPU-side BAM decode, see pd_push_bgzf_units */
    int (*push_bgzf_units)(pd_ctx *, const void *, size_t, const pd_bgzf_block *, uint32_t, const pd_bgzf_unit *, uint32_t,
                           uint64_t, uint32_t, int32_t, int32_t *, uint64_t *);
    /* optional (NULL = one context only): several GPUs in one process for `#.list` inputs */
    int (*device_count)(int *);
    int (*accumulate_from)(pd_ctx *dst, pd_ctx *src);
    /* optional (NULL = the host decodes): GPU-side BAM decode in asynchronous batches, see pd_decode_* */
    int (*decode_begin)(pd_ctx *, const pd_decode_cfg *);
    int (*decode_acquire)(pd_ctx *, size_t, void **);
    int (*decode_submit)(pd_ctx *, const pd_decode_batch *, int32_t *, pd_decode_result *);
    int (*decode_end)(pd_ctx *);
    int (*decode_abort)(pd_ctx *);
    int (*set_param)(pd_ctx *, const char *, uint64_t);
    /* optional (NULL = contexts are added into the first one): the RCCL sliced sum between one context per GPU, see pd_comm_* */
    int (*comm_init_all)(pd_ctx **, int, pd_comm **);
    int (*sliced_window_sum)(pd_comm *, uint32_t, uint32_t, unsigned, int, uint32_t *, uint64_t *);
    int (*comm_destroy)(pd_comm *);
    const char *(*comm_strerror)(const pd_comm *);
    /* optional (NULL = the host formats the cells it reads back): per-site rows formatted by the engine, see pd_format_sites */
    int (*format_sites)(pd_ctx *, int32_t, uint32_t, size_t, const char *, size_t, char *, size_t, size_t *);
    /* optional (NULL = every statistics call materialises the arrays): see pd_keep_deferred */
    int (*keep_deferred)(pd_ctx *, int);
    /* optional (NULL = zlib parses on the host threads): stage 1 of the byte-identical gzip streams on the engine, see pd_deflate_parse */
    int (*deflate_parse)(pd_ctx *, const void *, size_t, const pd_lz_chunk *, uint32_t, uint32_t *, size_t, uint64_t *);
    /* optional (NULL = buffers stay pageable): see pd_host_register */
    int (*host_register)(pd_ctx *, void *, size_t);
    int (*host_unregister)(pd_ctx *, void *);
    /* optional (NULL = the per-site text is produced on, or copied to, the host): the device-resident text stream, see pd_text_* */
    int (*text_open)(pd_ctx *, size_t, pd_text **);
    int (*text_close)(pd_text *);
    int (*text_append_sites)(pd_text *, int32_t, uint32_t, size_t, const char *, size_t, uint64_t *);
    int (*text_parse)(pd_text *, uint64_t, size_t, const pd_lz_chunk *, uint32_t, uint32_t *, size_t, uint64_t *, uint32_t *, uint64_t);
    int (*text_read)(pd_text *, uint64_t, size_t, void *);
    int (*text_release)(pd_text *, uint64_t);
    int (*text_append_window_rows)(pd_text *, int32_t, uint32_t, uint64_t, size_t, const char *, size_t, uint64_t *);
    int (*text_append_bytes)(pd_text *, const void *, size_t);
    /* optional (NULL = `#.list` inputs with -g / -b add the contexts into one GPU): see pd_sliced_interval_sum */
    int (*sliced_interval_sum)(pd_comm *, const pd_region *, size_t, uint32_t, unsigned, int, int32_t *, uint64_t *);
    /* optional (NULL = every batch is submitted and waited for by the thread that read it): the two halves of decode_submit, see pd_decode_queue */
    int (*decode_queue)(pd_ctx *, const pd_decode_batch *, uint64_t *);
    int (*decode_collect)(pd_ctx *, uint64_t, int32_t *, pd_decode_result *);
    /* optional (NULL = RCCL only): the in-process communicator over xGMI peer copies, see pd_comm_init_local; and RCCL loaded and
     * bootstrapped ahead of the contexts, see pd_comm_preinit */
    int (*comm_init_local)(pd_ctx **, int, pd_comm **);
    int (*comm_preinit)(const int *, int);
    /* optional (NULL = a communicator's exchange buffers are made by its first collective): see pd_comm_prepare */
    int (*comm_prepare)(pd_comm *, int);
} pd_engine_api;

/* Runs one `pandepth` invocation (argv as given to main) on the engine behind `api`. */
int pandepth_main(int argc, char **argv, const pd_engine_api *api, int device);

#ifdef __cplusplus
}
#endif
#endif
