// N native threads on ONE encoder handle, each embedding one short sentence per call through the host-pointer C ABI
// (kjarni_hip_encoder_embed_host) -- what a multi-threaded C# / Go / Rust caller of the library does, without a Python GIL in
// the way.  Prints calls/s and the latency distribution, with call combining on and off.
//   g++ -O2 -std=c++17 -pthread -o tools/lab/threads_bench tools/lab/threads_bench.cpp -ldl
//   tools/lab/threads_bench <libkjarni_ffi.so> <model_dir> [tokens=28] [calls per thread=400]
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

typedef int (*load_fn)(const char*, int32_t, void**);
typedef void (*free_fn)(void*);
typedef int (*embed_fn)(void*, const uint32_t*, const uint32_t*, const uint32_t*, int64_t, int32_t, int32_t, int32_t, int32_t, float*);
typedef int (*setc_fn)(void*, int32_t);
typedef int32_t (*hid_fn)(const void*);

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s libkjarni_ffi.so model_dir [tokens] [calls]\n", argv[0]);
        return 2;
    }
    const int tokens = argc > 3 ? std::atoi(argv[3]) : 28, calls = argc > 4 ? std::atoi(argv[4]) : 400;
    void* so = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!so) {
        std::fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    auto load = (load_fn)dlsym(so, "kjarni_hip_encoder_load");
    auto fre = (free_fn)dlsym(so, "kjarni_hip_encoder_free");
    auto embed = (embed_fn)dlsym(so, "kjarni_hip_encoder_embed_host");
    auto setc = (setc_fn)dlsym(so, "kjarni_hip_encoder_set_combining");
    auto hid = (hid_fn)dlsym(so, "kjarni_hip_encoder_hidden_size");
    void* enc = nullptr;
    if (!load || !embed || !setc || load(argv[2], 0, &enc) != 0) {
        std::fprintf(stderr, "load failed\n");
        return 1;
    }
    const int H = hid(enc);
    for (int combining = 1; combining >= 0; --combining) {
        setc(enc, combining);
        for (int threads : {1, 4, 16, 32}) {
            std::vector<std::vector<double>> lat(threads);
            std::atomic<int> failed{0};
            auto work = [&](int t) {
                std::vector<uint32_t> ids(tokens), mask(tokens, 1u);
                for (int i = 0; i < tokens; ++i) ids[i] = 1000u + (uint32_t)((t * 131 + i * 17) % 20000);
                ids[0] = 101;
                ids[tokens - 1] = 102;
                std::vector<float> out(H);
                for (int c = 0; c < calls + 20; ++c) {
                    const auto t0 = std::chrono::steady_clock::now();
                    if (embed(enc, ids.data(), mask.data(), nullptr, 1, tokens, 0, 1, 0, out.data()) != 0) failed++;
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    if (c >= 20) lat[t].push_back(us);
                }
            };
            const auto w0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            std::vector<double> all;
            for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
            std::sort(all.begin(), all.end());
            std::printf("{\"threads\": %d, \"combining\": %d, \"tokens\": %d, \"calls_per_s\": %.1f, \"p50_ms\": %.4f, \"p99_ms\": %.4f, \"failed\": %d}\n",
                        threads, combining, tokens, threads * (calls + 20) / wall, all[all.size() / 2] / 1e3, all[(size_t)(all.size() * 0.99)] / 1e3,
                        failed.load());
            std::fflush(stdout);
        }
    }
    fre(enc);
    return 0;
}
