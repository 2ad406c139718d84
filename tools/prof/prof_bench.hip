// prof_bench.hip -- torch-free driver of the C ABI for rocprofv3 (kernel stats and PMC
// passes at the bench size).  Generates the benchmark corpora of lz4_amd/corpus.py with a
// HIP kernel of its own (same bytes: every byte is a function of kind, seed and index), so
// that the profiler never sees torch's copy kernels, then runs the hot path through
// liblz4hip.so and prints one JSON line with HIP-event timings.
//
//   prof_bench TESTDATA_DIR KIND GIB BS_LOG WHAT [REPS] [LEVEL]
//     KIND  text | binary | mixed | random         WHAT  fast | decode | hc | xxh
// With a profiling build of the library on LD_LIBRARY_PATH (-DLZ4HIP_FAST_STATS, -DLZ4HIP_DEC_PROF,
// -DLZ4HIP_HC_PROF: cycle marks inside the kernels) the per-phase shares go to stderr as a second JSON
// line; LZ4HIP_PROF_RAW=1 prints the HC build's 16 slots as they are (experiment builds that use
// the slots their own way).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/lz4hip.h"

#define CK(call)                                                                       \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                 \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)
#define LZ(call)                                                                       \
    do {                                                                               \
        int r_ = (call);                                                               \
        if (r_ < 0) {                                                                  \
            fprintf(stderr, "%s: %d %s\n", #call, r_, lz4hip_last_error());            \
            exit(3);                                                                   \
        }                                                                              \
    } while (0)

static std::vector<uint8_t> slurp(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path.c_str());
        exit(1);
    }
    std::vector<uint8_t> v;
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

__device__ __forceinline__ uint32_t mix(uint64_t p, uint32_t seed) {  // corpus.py: _mix
    uint32_t x = (uint32_t)p ^ (uint32_t)((p >> 32) * 0x9E37ull) ^ seed;
    x = (x ^ (x >> 16)) * 0x45D9F3Bu;
    x = (x ^ (x >> 16)) * 0x45D9F3Bu;
    return x ^ (x >> 16);
}

// kind: 0 text, 1 binary, 2 random, 3 mixed (per 256 KiB region r: r%4 in {0,1} text, 2 binary, 3 random)
__global__ void k_corpus(uint8_t *out, size_t n, uint64_t start, int kind, uint32_t seed,
                         const uint8_t *text, uint32_t ntext, const uint8_t *bin, uint32_t nbin) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t p = start + i;
        const uint32_t h = mix(p, seed);
        int k = kind;
        if (kind == 3) {
            const uint32_t r = (uint32_t)((p >> 18) & 3);
            k = r < 2 ? 0 : (r == 2 ? 1 : 2);
        }
        uint8_t v;
        if (k == 2) {
            v = (uint8_t)h;
        } else if (k == 0) {
            v = text[p % ntext];
            if ((h & 63u) == 0) v = (uint8_t)(32 + ((h >> 8) % 95u));
        } else {
            v = bin[p % nbin];
            if ((h & 255u) == 0) v = (uint8_t)(h >> 8);
        }
        out[i] = v;
    }
}

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s TESTDATA KIND GIB BS_LOG WHAT [REPS] [LEVEL]\n", argv[0]);
        return 1;
    }
    const std::string dir = argv[1], kind_s = argv[2], what = argv[5];
    const double gib = atof(argv[3]);
    const uint32_t bs = 1u << atoi(argv[4]);
    const int reps = argc > 6 ? atoi(argv[6]) : 3;
    const uint32_t level = argc > 7 ? (uint32_t)atoi(argv[7]) : 131072;
    const int kind = kind_s == "text" ? 0 : kind_s == "binary" ? 1 : kind_s == "random" ? 2 : 3;
    const uint32_t seed = kind == 0 ? 0x5EED0002u : kind == 1 ? 0x5EED0004u : 0x5EED0003u;
    const size_t n = (size_t)(gib * (double)(1ull << 30)) / bs * bs;
    const size_t nb = n / bs;

    std::vector<uint8_t> text = slurp(dir + "/pg1661.txt");
    for (const char *f : {"/Mark.Twain-Tom.Sawyer.txt", "/gettysburg.txt"}) {
        auto v = slurp(dir + f);
        text.insert(text.end(), v.begin(), v.end());
    }
    std::vector<uint8_t> bin = slurp(dir + "/bzImage_lz4_isolated");
    {
        auto v = slurp(dir + "/pg_control.tar");
        bin.insert(bin.end(), v.begin(), v.end());
    }
    uint8_t *d_text, *d_bin, *d_src;
    CK(hipMalloc(&d_text, text.size()));
    CK(hipMalloc(&d_bin, bin.size()));
    CK(hipMalloc(&d_src, n + 16));
    CK(hipMemcpy(d_text, text.data(), text.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bin, bin.data(), bin.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_corpus, dim3(8192), dim3(256), 0, 0, d_src, n, 0ull, kind, seed, d_text,
                       (uint32_t)text.size(), d_bin, (uint32_t)bin.size());
    CK(hipDeviceSynchronize());

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](auto &&fn) {
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0, 0));
            fn();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            sum += ms;
        }
        return std::make_pair(best, sum / reps);
    };

    uint8_t *d_comp;
    int32_t *d_res;
    CK(hipMalloc(&d_comp, nb * (size_t)bs));
    CK(hipMalloc(&d_res, (nb + 1) * 4));
    uint32_t *d_xxh;
    CK(hipMalloc(&d_xxh, (nb + 1) * 4));

    if (what == "xxh") {
        auto t = timed([&] { LZ(lz4hip_xxh32_blocks(d_src, n, bs, nullptr, nullptr, nb, d_xxh, nullptr)); });
        printf("{\"what\":\"xxh32_blocks\",\"kind\":\"%s\",\"n\":%zu,\"bs\":%u,\"ms_min\":%.3f,\"ms_avg\":%.3f,\"GBps\":%.2f}\n",
               kind_s.c_str(), n, bs, t.first, t.second, n / (t.first * 1e6));
        return 0;
    }

    void *d_ws = nullptr;
    size_t ws_bytes = 0;
    const bool hc = what == "hc";
    ws_bytes = hc ? lz4hip_hc_workspace_bytes(nb) : lz4hip_fast_workspace_bytes(nb);
    if (const char *e = getenv("LZ4_PROF_SLOTS"))  // resident waves of the Fast kernel (wave sweep)
        if (!hc && atoi(e) > 0) ws_bytes = (size_t)atoi(e) * 131072 + 256;
    if (ws_bytes) {
        CK(hipMalloc(&d_ws, ws_bytes));
        CK(hipMemset((uint8_t *)d_ws + ws_bytes - 256, 0, 256));
    }
    auto compress = [&] {
        if (hc)
            LZ(lz4hip_compress_hc_blocks(d_src, n, bs, level, LZ4HIP_DST_FRAME, d_comp, bs, d_res, d_ws,
                                         ws_bytes, nullptr));
        else
            LZ(lz4hip_compress_fast_blocks(d_src, n, bs, LZ4HIP_DST_FRAME, d_comp, bs, d_res, d_ws, ws_bytes,
                                           nullptr));
    };
    std::pair<float, float> tc(0, 0);
    if (what == "decode") {
        compress();  // set-up for the decode input (shows up as one extra dispatch in a trace)
        CK(hipDeviceSynchronize());
    } else {
        tc = timed(compress);
    }
    std::vector<int32_t> res(nb);
    CK(hipMemcpy(res.data(), d_res, nb * 4, hipMemcpyDeviceToHost));
    size_t c = 0, raw = 0;
    for (size_t b = 0; b < nb; b++) {
        if (res[b] > 0) c += (size_t)res[b];
        else c += bs, raw++;
    }
    if (what == "fast" && ws_bytes >= 256) {  // -DLZ4HIP_FAST_STATS builds: access counts by class
        unsigned long long st[16];
        CK(hipMemcpy(st, (uint8_t *)d_ws + ws_bytes - 256, sizeof st, hipMemcpyDeviceToHost));
        if (st[7]) {
            const double per = (double)st[9];  // bytes over all timed repetitions
            fprintf(stderr,
                    "{\"fast_stats_per_input_byte\":{\"table_gathers\":%.4f,\"candidate_verify\":%.4f,\"ext_r0\":%.4f,"
                    "\"ext_r1\":%.4f,\"ext_rb\":%.4f,\"table_puts\":%.4f,\"refill_lines\":%.4f,\"bytes_per_window\":%.2f}}\n",
                    st[1] / per, st[2] / per, st[3] / per, st[4] / per, st[5] / per, st[6] / per, st[8] / per,
                    per / st[7]);
            const double tot = (double)(st[10] + st[11] + st[12] + st[13] + st[14] + st[15]);
            if (tot > 0)
                fprintf(stderr,
                        "{\"fast_phase_share\":{\"refill\":%.3f,\"table_gather\":%.3f,\"verify_ext\":%.3f,\"walk\":%.3f,"
                        "\"encode\":%.3f,\"puts\":%.3f},\"cycles_per_window\":%.0f}\n",
                        st[10] / tot, st[11] / tot, st[12] / tot, st[13] / tot, st[14] / tot, st[15] / tot, tot / st[7]);
        }
    }
    if (what == "fast") {  // -DLZ4HIP_TEAM_PROF builds: cycles per role and phase of the team kernel
        typedef int (*prof_fn)(unsigned long long *);
        if (prof_fn f = (prof_fn)dlsym(RTLD_DEFAULT, "lz4hip_debug_team_prof")) {
            unsigned long long st[16];
            if (f(st) == 0 && st[4]) {
                const double w = (double)st[4];
                fprintf(stderr,
                        "{\"team_cycles_per_walker_window\":{\"walker\":{\"wait_window\":%.0f,\"records_stale\":%.0f,\"walk\":%.0f,"
                        "\"puts_handover\":%.0f},\"producers_per_own_window\":{\"wait\":%.0f,\"fill\":%.0f,\"phase_a\":%.0f},"
                        "\"encoder_per_record\":{\"wait\":%.0f,\"encode\":%.0f}},\"walker_windows\":%llu,\"stale_windows\":%llu,"
                        "\"records\":%llu,\"produced_windows\":%llu,\"rare_steps\":%llu,\"bytes_per_window\":%.1f}\n",
                        st[0] / w, st[1] / w, st[2] / w, st[3] / w, st[8] / (double)(st[11] ? st[11] : 1),
                        st[9] / (double)(st[11] ? st[11] : 1), st[10] / (double)(st[11] ? st[11] : 1),
                        st[12] / (double)(st[14] ? st[14] : 1), st[13] / (double)(st[14] ? st[14] : 1), st[4], st[5],
                        st[6], st[11], st[7], (double)n * reps / w);
            }
        }
    }
    if (what == "hc") {  // -DLZ4HIP_HC_PROF builds of the library export per-phase cycle sums
        typedef int (*prof_fn)(unsigned long long *);
        if (prof_fn f = (prof_fn)dlsym(RTLD_DEFAULT, "lz4hip_debug_hc_prof")) {
            unsigned long long st[16];
            if (f(st) == 0 && getenv("LZ4HIP_PROF_RAW")) {  // experiment builds that use the slots their own way
                fprintf(stderr, "{\"hc_raw\":[");
                for (int k = 0; k < 16; k++) fprintf(stderr, "%llu%s", st[k], k < 15 ? "," : "]}\n");
            } else if (f(st) == 0 && st[8] + st[9] + st[10] + st[11]) {
                double tot = 0;
                for (int i = 0; i < 8; i++) tot += (double)st[i];
                const double ns = (double)(st[8] + st[9] + st[10] + st[11]);
                fprintf(stderr,
                        "{\"hc_time_share\":{\"search_empty\":%.3f,\"search_1_batch\":%.3f,\"search_2_8\":%.3f,"
                        "\"search_more\":%.3f,\"store_bookkeeping\":%.3f,\"insertion\":%.3f,\"emit\":%.3f,\"setup_tail\":%.3f},"
                        "\"search_share\":{\"empty\":%.3f,\"one\":%.3f,\"2_8\":%.4f,\"more\":%.4f},"
                        "\"batches_per_search\":%.2f,\"matches_per_search\":%.3f,\"cycles_per_search\":%.0f}\n",
                        st[0] / tot, st[1] / tot, st[2] / tot, st[3] / tot, st[4] / tot, st[5] / tot, st[6] / tot,
                        st[7] / tot, st[8] / ns, st[9] / ns, st[10] / ns, st[11] / ns, st[12] / ns, st[13] / ns,
                        tot / ns);
            }
        }
    }
    if (what != "decode") {
        printf("{\"what\":\"%s\",\"kind\":\"%s\",\"n\":%zu,\"bs\":%u,\"blocks\":%zu,\"raw_blocks\":%zu,\"c\":%zu,"
               "\"ms_min\":%.3f,\"ms_avg\":%.3f,\"GBps\":%.3f,\"alg_GBps\":%.3f,\"roofline_frac\":%.5f}\n",
               what.c_str(), kind_s.c_str(), n, bs, nb, raw, c, tc.first, tc.second, n / (tc.first * 1e6),
               (n + c + 8 * nb) / (tc.first * 1e6), (n + c + 8 * nb) / (tc.first * 1e6) / 8000.0);
        return 0;
    }
    // decode: pack the blocks like a frame body, then time lz4hip_uncompress_blocks
    uint64_t *d_off;
    uint32_t *d_sw;
    CK(hipMalloc(&d_off, (nb + 1) * 8));
    CK(hipMalloc(&d_sw, (nb + 1) * 4));
    LZ(lz4hip_frame_layout(d_res, n, bs, nb, nullptr, 0, d_off, d_sw, nullptr));
    uint64_t end = 0;
    CK(hipMemcpy(&end, d_off + nb, 8, hipMemcpyDeviceToHost));
    uint8_t *d_packed;
    CK(hipMalloc(&d_packed, end + 16));
    LZ(lz4hip_frame_scatter(d_src, n, bs, d_comp, bs, d_res, nullptr, d_off, nb, d_packed, 0, nullptr));
    std::vector<uint64_t> off(nb);
    CK(hipMemcpy(off.data(), d_off, nb * 8, hipMemcpyDeviceToHost));
    for (auto &o : off) o += 4;
    CK(hipMemcpy(d_off, off.data(), nb * 8, hipMemcpyHostToDevice));
    int32_t *d_len;
    CK(hipMalloc(&d_len, nb * 4));
    uint8_t *d_out = d_comp;  // reuse: nb * bs bytes
    auto td = timed([&] {
        LZ(lz4hip_uncompress_blocks(d_packed, end, d_off, d_sw, nb, d_out, bs, bs, nullptr, 0, d_len, nullptr));
    });
    // parity: the decoded bytes are the corpus (checksum of checksums on both sides)
    uint32_t *d_x2;
    CK(hipMalloc(&d_x2, nb * 4));
    LZ(lz4hip_xxh32_blocks(d_src, n, bs, nullptr, nullptr, nb, d_xxh, nullptr));
    LZ(lz4hip_xxh32_blocks(d_out, n, bs, nullptr, d_len, nb, d_x2, nullptr));
    std::vector<uint32_t> x1(nb), x2(nb);
    CK(hipMemcpy(x1.data(), d_xxh, nb * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(x2.data(), d_x2, nb * 4, hipMemcpyDeviceToHost));
    const bool ok = x1 == x2;
    const size_t cc = end - 4 * nb;
    {  // -DLZ4HIP_DEC_PROF builds of the library export per-phase cycle sums
        typedef int (*prof_fn)(unsigned long long *);
        if (prof_fn f = (prof_fn)dlsym(RTLD_DEFAULT, "lz4hip_debug_dec_prof")) {
            unsigned long long st[16];
            if (f(st) == 0 && st[7]) {
                double tot = 0;
                for (int i = 0; i < 12; i++) if (i != 7) tot += (double)st[i];
                fprintf(stderr, "{\"copies_split\":{\"literals\":%.3f,\"match_fetch\":%.3f,\"match_store\":%.3f}}\n", st[9] / tot, st[10] / tot, st[4] / tot);
                fprintf(stderr,
                        "{\"decode_phase_share\":{\"refill\":%.3f,\"pass1\":%.3f,\"link\":%.3f,\"compact_parse_scan\":%.3f,"
                        "\"copies\":%.3f,\"in_window\":%.3f,\"flush\":%.3f,\"other\":%.3f},\"cycles_per_window\":%.0f}\n",
                        st[0] / tot, st[1] / tot, st[2] / tot, st[3] / tot, st[4] / tot, st[5] / tot, st[6] / tot,
                        st[8] / tot, tot / st[7]);
            }
        }
    }
    printf("{\"what\":\"decode\",\"kind\":\"%s\",\"n\":%zu,\"bs\":%u,\"blocks\":%zu,\"c\":%zu,\"ms_min\":%.3f,"
           "\"ms_avg\":%.3f,\"GBps\":%.3f,\"alg_GBps\":%.3f,\"roofline_frac\":%.5f,\"round_trip_ok\":%s}\n",
           kind_s.c_str(), n, bs, nb, cc, td.first, td.second, n / (td.first * 1e6), (n + cc) / (td.first * 1e6),
           (n + cc) / (td.first * 1e6) / 8000.0, ok ? "true" : "false");
    return ok ? 0 : 4;
}
