// main.cpp — the `pandepth` executable: the reference's command line on the MI355X depth engine.
// The engine table is bound to libpandepth_amd.so's entry points and to nothing else; without a
// gfx950 device pd_create fails and the program exits with an error (no CPU fallback).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>
#include "engine_api.h"

// PANDEPTH_TIMING: wall-clock stamps at both ends of main, so that a caller that notes the clock around the process sees what lies
// outside (the loader before, the kernel's teardown of the process's device and pinned memory after)
static void stamp(const char *what)
{
    if (!getenv("PANDEPTH_TIMING")) return;
    timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
    fprintf(stderr, "[timing] %s at %lld.%06ld (epoch)\n", what, (long long)ts.tv_sec, ts.tv_nsec / 1000);
}

int main(int argc, char **argv)
{
    stamp("main entered");
    static const pd_engine_api api = {
        pd_create, pd_destroy, pd_strerror, pd_push_intervals, pd_scan, pd_reduce_intervals,
        pd_window_layout, pd_scan_reduce_windows, pd_reduce_windows, pd_read_depth, pd_synchronize, pd_push_bgzf_units, pd_device_count, pd_accumulate_from,
        pd_decode_begin, pd_decode_acquire, pd_decode_submit, pd_decode_end, pd_decode_abort, pd_set_param,
        pd_comm_init_all, pd_sliced_window_sum, pd_comm_destroy, pd_comm_strerror, pd_format_sites, pd_keep_deferred, pd_deflate_parse, pd_host_register, pd_host_unregister,
        pd_text_open, pd_text_close, pd_text_append_sites, pd_text_parse, pd_text_read, pd_text_release, pd_text_append_window_rows, pd_text_append_bytes, pd_sliced_interval_sum,
        pd_decode_queue, pd_decode_collect, pd_comm_init_local, pd_comm_preinit, pd_comm_prepare,
    };
    // The decoder keeps six batches in flight on six streams (plus the context's main stream, which carries the batches' copies, and the compose stream:
    // eight); the runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues — 4 unless told otherwise — and streams that share a queue wait
    // for each other's kernels AND for each other's markers (profiles/r06_devtrace.txt: with nine streams on eight queues a finished batch was collected
    // 2 ms late).  Eight: every stream of a whole-contig run its own queue (a queue is made when its stream first launches: 9 ms each); sixteen measured
    // the same there and 3 % slower on the per-site writer (tools/calls/r6_call21.sh).
    {   // (a `#.list` input: every context keeps four readers x two buffers — eight slot streams + the main and the compose stream per GPU: sixteen there)
        bool list = false;
        for (int a = 1; a < argc; ++a) { const char *x = argv[a]; const size_t n = strlen(x); if (n > 5 && !strcmp(x + n - 5, ".list")) list = true; }
        setenv("GPU_MAX_HW_QUEUES", list ? "16" : "8", 0);
    }
    const char *dev = getenv("PANDEPTH_DEVICE");
    // every output file is closed when pandepth_main returns; freeing tens of GB of HBM and unloading the HIP runtime in
    // order would only delay the exit (0.1-0.2 s), so the process ends here and the driver reclaims the device memory
    // (PANDEPTH_ORDERLY_EXIT=1: tear everything down and return — a profiler attached to the process writes its files from an
    // exit handler, which _exit would skip)
    const bool orderly = getenv("PANDEPTH_ORDERLY_EXIT") != nullptr;
    if (!orderly) setenv("PANDEPTH_KEEP_CONTEXT", "1", 1);
    const int rc = pandepth_main(argc, argv, &api, dev ? atoi(dev) : 0);
    if (getenv("PANDEPTH_GUARD") && pd_guard_check(nullptr, 0) > 0) {          // debugging aid: a kernel wrote outside one of the engine's buffers
        fprintf(stderr, "pandepth: PANDEPTH_GUARD found an out-of-bounds device write (see above)\n");
        fflush(stderr);
        _exit(97);
    }
    stamp("main leaving");
    fflush(stdout); fflush(stderr);
    if (orderly) return rc;
    _exit(rc);
}
