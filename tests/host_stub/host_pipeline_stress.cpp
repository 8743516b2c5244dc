// host_pipeline_stress.cpp -- drives the host pipeline of csrc/capi.hip (built with -DBZ_HOST_PIPELINE_TEST against
// stub_engine.cpp and hip_shim.h) under a sanitizer: random write sizes, Actions, device lists and chunk sizes;
// contexts side by side on several threads; contexts destroyed while jobs are in flight; the resource cache released
// while other contexts run.  What must hold:
//   * the bytes of a stream do not depend on the device list, the chunk size or the write sizes -- every stream is
//     compared with the same (pieces, Actions) sequence through ONE lane pair with chunks larger than the input;
//   * a Finish-only stream decodes (the stub's framing) to exactly the input;
//   * no data race, no use after free, no leak (the sanitizer's business).
// usage: host_pipeline_stress [iterations] [seed]
#include "../../include/bz2_mi355x.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

typedef std::vector<uint8_t> Bytes;
struct Step {
    size_t n;   // bytes written before the action
    int action; // BZ_ACTION_*
    bool twice; // the action once more with no input in between
};

static Bytes make_input(std::mt19937_64 &rng, size_t n)
{
    Bytes d(n);
    size_t i = 0;
    while (i < n) {
        const unsigned kind = rng() % 8;
        size_t k = 1 + rng() % 4000;
        if (kind == 0) k = 1 + rng() % 70000; // long runs: the pending-run logic of Flush / Run
        if (k > n - i) k = n - i;
        if (kind <= 1) memset(&d[i], (int)(rng() % 3), k);
        else for (size_t j = 0; j < k; ++j) d[i + j] = (uint8_t)(rng() % 251);
        i += k;
    }
    return d;
}

static int run_stream(const Bytes &data, const std::vector<Step> &steps, int level, const std::vector<int> &devices, size_t piece,
                      Bytes &out)
{
    bz_enc *e = nullptr;
    int rc = bz_enc_create_multi(&e, level, devices.data(), (int)devices.size());
    if (rc != BZ_OK) return rc;
    size_t pos = 0;
    uint8_t buf[65536];
    out.clear();
    for (const Step &s : steps) {
        size_t left = s.n;
        while (left && rc == BZ_OK) {
            const size_t k = left < piece ? left : piece;
            rc = bz_enc_write(e, data.data() + pos, k);
            pos += k;
            left -= k;
            long got;
            while ((got = bz_enc_read(e, buf, sizeof(buf))) > 0) out.insert(out.end(), buf, buf + got);
        }
        for (int t = 0; t < (s.twice ? 2 : 1) && rc == BZ_OK; ++t) {
            rc = bz_enc_end(e, s.action);
            long got;
            while ((got = bz_enc_read(e, buf, sizeof(buf))) > 0) out.insert(out.end(), buf, buf + got);
        }
        if (rc != BZ_OK) break;
    }
    bz_enc_destroy(e);
    return rc;
}

// the stub's framing, Finish-only streams: header, blocks (magic 16 | length 24 | checksum 32 | bytes | 101), trailer
static bool decode_stub(const Bytes &z, Bytes &back)
{
    uint64_t pos = 0;
    const uint64_t nbits = (uint64_t)z.size() * 8;
    auto get = [&](unsigned k) {
        uint64_t v = 0;
        for (unsigned i = 0; i < k; ++i, ++pos) v = (v << 1) | (pos < nbits ? (z[pos >> 3] >> (7 - (pos & 7))) & 1u : 0u);
        return v;
    };
    back.clear();
    if ((get(32) & 0xFFFFFFF0u) != 0x425A6830u) return false;
    for (;;) {
        const uint64_t m = get(16);
        if (m == 0xB10C) {
            const uint64_t len = get(24);
            (void)get(32);
            for (uint64_t i = 0; i < len; ++i) back.push_back((uint8_t)get(8));
            if (get(3) != 5) return false;
        } else {
            pos -= 16;
            if (get(48) != 0x177245385090ull) return false;
            (void)get(32);
            return (nbits - pos) < 8;
        }
        if (pos > nbits) return false;
    }
}

static int one_case(uint64_t seed, bool verbose)
{
    std::mt19937_64 rng(seed);
    const size_t n = (rng() % 4 == 0) ? rng() % 3000 : 20000 + rng() % 1500000;
    const Bytes data = make_input(rng, n);
    const int level = 1 + (int)(rng() % 9);
    std::vector<Step> steps;
    size_t left = n;
    const bool finish_only = rng() % 3 == 0;
    while (true) {
        Step s;
        s.n = finish_only ? left : (rng() % 4 == 0 ? 0 : rng() % (left + 1));
        if (rng() % 5 == 0) s.n = left;
        left -= s.n;
        s.action = left == 0 && (finish_only || rng() % 2) ? BZ_ACTION_FINISH : (int)(rng() % 2); // Run / Flush in between
        s.twice = rng() % 5 == 0;
        steps.push_back(s);
        if (s.action == BZ_ACTION_FINISH) break;
        if (steps.size() > 12) { // end it
            steps.push_back({left, BZ_ACTION_FINISH, false});
            left = 0;
            break;
        }
    }
    // the reference: one device, chunks larger than the input, the same pieces and Actions
    Bytes ref, got, back;
    setenv("BZ_ENC_CHUNK_BYTES", "1073741824", 1);
    int rc = run_stream(data, steps, level, {0}, n + 1, ref);
    if (rc != BZ_OK) { fprintf(stderr, "seed %llu: reference run failed: %d\n", (unsigned long long)seed, rc); return 1; }
    if (finish_only) {
        if (!decode_stub(ref, back) || back != data) { fprintf(stderr, "seed %llu: the reference stream does not decode to the input\n", (unsigned long long)seed); return 1; }
    }
    static const std::vector<std::vector<int>> lists = {{0}, {1, 0}, {0, 0, 0}, {0, 1, 2, 3}, {2, 2}};
    static const size_t chunks[] = {4096, 20000, 65536, 300000};
    for (int v = 0; v < 3; ++v) {
        const std::vector<int> &devs = lists[rng() % lists.size()];
        const size_t chunk = chunks[rng() % 4];
        const size_t piece = 1 + rng() % (rng() % 2 ? 5000 : 400000);
        setenv("BZ_ENC_CHUNK_BYTES", std::to_string(chunk).c_str(), 1);
        rc = run_stream(data, steps, level, devs, piece, got);
        if (rc != BZ_OK || got != ref) {
            size_t d = 0;
            while (d < got.size() && d < ref.size() && got[d] == ref[d]) ++d;
            fprintf(stderr, "first difference at byte %zu; steps:", d);
            for (const Step &s : steps) fprintf(stderr, " (%zu, action %d%s)", s.n, s.action, s.twice ? " x2" : "");
            fprintf(stderr, "\n");
            fprintf(stderr, "seed %llu: %zu bytes, level %d, %zu steps, %zu devices, chunk %zu, pieces of %zu: rc %d, %zu bytes against %zu\n",
                    (unsigned long long)seed, n, level, steps.size(), devs.size(), chunk, piece, rc, got.size(), ref.size());
            return 1;
        }
    }
    // one-shot over a device list
    {
        const std::vector<int> &devs = lists[rng() % lists.size()];
        setenv("BZ_ENC_CHUNK_BYTES", "30000", 1);
        uint8_t *o = nullptr;
        size_t on = 0;
        rc = bz_encode_buffer_multi(level, devs.data(), (int)devs.size(), data.data(), data.size(), &o, &on);
        Bytes z(o, o + on);
        bz_free(o);
        if (rc != BZ_OK || !decode_stub(z, back) || back != data) {
            fprintf(stderr, "seed %llu: one-shot over %zu devices: rc %d, decodes %d\n", (unsigned long long)seed, devs.size(), rc, (int)(back == data));
            return 1;
        }
    }
    if (verbose) fprintf(stderr, "seed %llu ok (%zu bytes, %zu steps)\n", (unsigned long long)seed, n, steps.size());
    return 0;
}

// contexts that are destroyed while their jobs are in flight, on several threads, while another thread releases the cache
static int destroy_while_busy(uint64_t seed)
{
    std::mt19937_64 rng(seed);
    setenv("BZ_ENC_CHUNK_BYTES", "16384", 1);
    const Bytes data = make_input(rng, 600000);
    std::vector<std::thread> th;
    for (int t = 0; t < 3; ++t)
        th.emplace_back([&data, t] {
            for (int it = 0; it < 6; ++it) {
                bz_enc *e = nullptr;
                const int devs[3] = {t, (t + 1) % 4, t};
                if (bz_enc_create_multi(&e, 5, devs, 1 + (it % 3)) != BZ_OK) continue;
                (void)bz_enc_write(e, data.data(), data.size() / (1 + it % 3));
                if (it % 2) (void)bz_enc_end(e, BZ_ACTION_FLUSH);
                bz_enc_destroy(e); // (jobs may still be queued or running)
            }
        });
    th.emplace_back([] {
        for (int it = 0; it < 10; ++it) {
            bz_release_cached_resources();
            std::this_thread::yield();
        }
    });
    for (auto &t : th) t.join();
    bz_release_cached_resources();
    return 0;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const uint64_t seed0 = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    int bad = 0;
    for (int i = 0; i < iters && !bad; ++i) bad |= one_case(seed0 + (uint64_t)i, argc > 3);
    if (!bad) bad |= destroy_while_busy(seed0);
    // several streams side by side (contexts share the resource cache and the shim's devices); the references are made
    // first, one after the other, under the same settings
    if (!bad) {
        setenv("BZ_ENC_CHUNK_BYTES", "20000", 1);
        std::mt19937_64 rng(seed0 * 7919u);
        std::vector<Bytes> inputs, refs(4);
        for (int t = 0; t < 4; ++t) inputs.push_back(make_input(rng, 200000 + (size_t)t * 77777));
        const std::vector<Step> plan = {{50000, BZ_ACTION_RUN, false}, {60001, BZ_ACTION_FLUSH, true}, {0, BZ_ACTION_RUN, false}};
        auto steps_for = [&](size_t n) {
            std::vector<Step> s = plan;
            s.push_back({n - 110001, BZ_ACTION_FINISH, false});
            return s;
        };
        for (int t = 0; t < 4; ++t) bad |= run_stream(inputs[(size_t)t], steps_for(inputs[(size_t)t].size()), 3, {0}, 33333, refs[(size_t)t]) != BZ_OK;
        std::vector<std::thread> th;
        std::vector<int> res(4, 0);
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                static const std::vector<std::vector<int>> lists = {{0, 1}, {1, 1, 1}, {3}, {2, 0, 1, 3}};
                for (int it = 0; it < 3; ++it) {
                    Bytes got;
                    const int rc = run_stream(inputs[(size_t)t], steps_for(inputs[(size_t)t].size()), 3, lists[(size_t)((t + it) % 4)],
                                              1000 + 7000 * (size_t)it, got);
                    if (rc != BZ_OK || got != refs[(size_t)t]) res[(size_t)t] = 1;
                }
            });
        for (auto &t : th) t.join();
        for (int r : res) bad |= r;
        if (bad) fprintf(stderr, "streams side by side: a stream differs from its reference\n");
    }
    if (!bad) { // the multi-device diagnostic (bench.py's preflight): neighbour pairs of a device list, one of them on one device
        const int devs[4] = {0, 1, 1, 3};
        int peer[4] = {9, 9, 9, 9};
        double ms[4] = {-2, -2, -2, -2};
        const int rc = bz_peer_copy_selftest(devs, 4, 100000, peer, ms);
        if (rc != BZ_OK || peer[0] != 1 || peer[1] != -1 || peer[2] != 1 || peer[3] != 1 || ms[0] < 0 || ms[1] != 0.0 || ms[3] < 0 ||
            bz_peer_copy_selftest(devs, 0, 1, nullptr, nullptr) != BZ_E_PARAM || bz_peer_copy_selftest(nullptr, 2, 1, nullptr, nullptr) != BZ_E_PARAM) {
            fprintf(stderr, "bz_peer_copy_selftest: rc %d, peer %d %d %d %d\n", rc, peer[0], peer[1], peer[2], peer[3]);
            bad = 1;
        }
    }
    bz_release_cached_resources();
    printf(bad ? "FAILED\n" : "ok\n");
    return bad;
}
