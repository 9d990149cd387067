// Test-only driver: the library's multi-threaded HOST code under ThreadSanitizer (CPU build; GPU sanitizers are not
// available on the pool).  What runs on more than one thread beside the kernels: the level-wise parallel walk
// (walk.cpp), the ranged YAML emitter and parser (hostpass.cpp), the persistent pool of staging-fill threads (hostfill.cpp
// FillPool), the read-ahead reader of one long host-hashed file (hostsha.cpp) and the thread sets of a planned host
// part (ThreadJoiner), and -- round 5 -- the host threads that hash the fused Build pass's long members out of the staging
// slots (member_hashers.h).  Exit code 0 = no report, every result equal to the single-threaded one.
//
// usage: tsan_host BUILD_DIR BIG_FILE
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include "../snappy_amd/csrc/hostfill.h"
#include "../snappy_amd/csrc/hostpass.h"
#include "../snappy_amd/csrc/hostsha.h"
#include "../snappy_amd/csrc/member_hashers.h"

using namespace snaphash;

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    // 1. the walk, several times over (its worker sets are created per call), and the emitter over its records
    std::vector<Record> first;
    std::string first_yaml;
    for (int round = 0; round < 4; ++round) {
        std::vector<Record> recs;
        int en = 0;
        if (walk_tree(argv[1], recs, &en) != SNAPHASH_OK) return 3;
        if (recs.size() < 4096) return 4; // below that neither the walk nor the emitter goes parallel
        std::vector<uint8_t> dig(64 * (recs.size() + 1));
        for (size_t i = 0; i < dig.size(); ++i) dig[i] = (uint8_t)(i * 131u >> 3);
        std::string y;
        if (emit_yaml(recs, dig.data(), dig.data() + 64, y) != SNAPHASH_OK) return 5;
        { // round 5: the document written ahead of its digests on a background thread while this one walks again (the helper
          // pool is busy then: the walk falls back to threads of its own), the digests filled in afterwards -- the same bytes
            YamlSkeleton sk;
            int sk_rc = -1;
            std::vector<Record> again;
            {
                ThreadJoiner bg;
                bg.spawn([&] { sk_rc = emit_yaml_skeleton(recs, sk, round % 2 ? 1 : 8); });
                int en2 = 0;
                if (walk_tree(argv[1], again, &en2) != SNAPHASH_OK) return 20;
                bg.join_all();
            }
            if (sk_rc != SNAPHASH_OK || sk.hex_at.size() != 1 + (size_t)std::count_if(recs.begin(), recs.end(), [](const Record& r) { return r.is_regular; })) return 21;
            if (again.size() != recs.size()) return 22;
            yaml_fill_digests(sk, dig.data(), dig.data() + 64);
            if (sk.text != y) return 23;
            if (first_unemittable_name(recs) != recs.size()) return 24;
        }
        if (round == 0) {
            first = recs;
            first_yaml = y;
            ParsedHashes ph; // (40 000 lines and more: parsed in ranges, a thread each)
            if (parse_yaml(y.data(), y.size(), ph) != SNAPHASH_OK || ph.files.size() != recs.size()) return 6;
            for (size_t i = 0; i < recs.size(); ++i)
                if (ph.files[i].name != recs[i].name) return 7;
        } else {
            if (recs.size() != first.size() || y != first_yaml) return 8;
            for (size_t i = 0; i < recs.size(); ++i)
                if (recs[i].name != first[i].name || recs[i].size != first[i].size || recs[i].st_mode != first[i].st_mode) return 9;
        }
    }

    // 2. the fill pool: many short jobs back to back (the epoch hand-over is where a lost wake-up or a stale fn_
    //    would show), resized in between, destroyed with threads parked
    {
        FillPool pool;
        pool.configure(6, {});
        std::vector<uint32_t> slots(5000);
        for (int job = 0; job < 300; ++job) {
            const size_t n = 1 + (size_t)(job * 37 % 5000);
            std::atomic<size_t> calls{0};
            const std::function<void(size_t)> fn = [&](size_t i) {
                slots[i] = (uint32_t)(i * 2654435761u) + (uint32_t)job;
                calls.fetch_add(1, std::memory_order_relaxed);
            };
            pool.parallel_for(n, 1 + job % 7, fn);
            if (calls.load() != n) return 10;
            for (size_t i = 0; i < n; ++i)
                if (slots[i] != (uint32_t)(i * 2654435761u) + (uint32_t)job) return 11;
            if (job == 150) pool.configure(3, {});
            if (job == 220) pool.configure(8, {});
        }
        // copies into a buffer the pool's threads share, disjoint ranges, non-temporal path included
        std::vector<uint8_t> src(8u << 20), dst(8u << 20);
        for (size_t i = 0; i < src.size(); ++i) src[i] = (uint8_t)(i * 7u + (i >> 11));
        const std::function<void(size_t)> cp = [&](size_t i) { copy_to_staging(dst.data() + i * (256u << 10), src.data() + i * (256u << 10), 256u << 10); };
        pool.parallel_for(32, 6, cp);
        if (memcmp(src.data(), dst.data(), src.size()) != 0) return 12;
    }

    // 3. one long file on a host thread with the reader thread ahead of it, against the plain loop
    {
        uint8_t a[64], b[64];
        HostSha s1, s2;
        host_sha512_init(s1);
        host_sha512_init(s2);
        FILE* f = fopen(argv[2], "rb");
        if (!f) return 13;
        fseek(f, 0, SEEK_END);
        const uint64_t len = (uint64_t)ftell(f);
        fclose(f);
        if (len < (32u << 20)) return 14; // the read-ahead starts at 32 MiB
        if (host_sha512_file_from(s1, argv[2], 0, len, a, false) != 0) return 15;
        if (host_sha512_file_from(s2, argv[2], 0, len, b, true) != 0) return 16;
        if (memcmp(a, b, 64) != 0) return 17;
        // and two of them at once, as two host threads of a planned call would run them
        uint8_t c[64], d[64];
        int rc1 = -1, rc2 = -1;
        {
            ThreadJoiner tj;
            tj.th.emplace_back([&] { HostSha s; host_sha512_init(s); rc1 = host_sha512_file_from(s, argv[2], 0, len, c, true); });
            tj.th.emplace_back([&] { HostSha s; host_sha512_init(s); rc2 = host_sha512_file_from(s, argv[2], 0, len, d, true); });
        }
        if (rc1 != 0 || rc2 != 0 || memcmp(a, c, 64) != 0 || memcmp(a, d, 64) != 0) return 18;
        // a file that is shorter than the caller says: the reader thread must stop and be joined, an error come back
        HostSha s3;
        host_sha512_init(s3);
        if (host_sha512_file_from(s3, argv[2], 0, len + 4096, a, true) == 0) return 19;
    }
    // 4. a worker that throws (an allocation failing inside a thread): contained there, raised on the owner's thread, and
    //    the pool goes on working
    {
        bool raised = false;
        try {
            ThreadJoiner tj;
            tj.spawn([](int k) { if (k == 1) throw std::bad_alloc(); }, 1);
            tj.spawn([](int) {}, 0);
            tj.join_all();
        } catch (const std::bad_alloc&) {
            raised = true;
        }
        if (!raised) return 20;
        FillPool pool;
        pool.configure(4, {});
        for (int where = 0; where < 2; ++where) { // thrown on a pool thread (some item far from the caller's first), then on any
            raised = false;
            std::atomic<size_t> done{0};
            const std::function<void(size_t)> fn = [&](size_t i) {
                if (where == 0 ? i == 777 : i % 97 == 5) throw std::bad_alloc();
                done.fetch_add(1, std::memory_order_relaxed);
            };
            try { pool.parallel_for(2000, 4, fn); } catch (const std::bad_alloc&) { raised = true; }
            if (!raised) return 21;
        }
        std::atomic<size_t> ok{0};
        const std::function<void(size_t)> fine = [&](size_t) { ok.fetch_add(1, std::memory_order_relaxed); };
        pool.parallel_for(3000, 4, fine);
        if (ok.load() != 3000) return 22;
    }
    // 6. round 5: the fused Build pass's long members on host threads (member_hashers.h).  A "packer" refills two slot buffers
    //    in turn with the next stretch of a stream of members -- waiting for the workers that still read the buffer it is about
    //    to overwrite -- and hands every member's piece to ITS worker; a member's digest must be the single-shot one.
    {
        const size_t kSlot = 192 << 10;
        std::vector<uint64_t> sizes = {700001, 5, 200000, 1 << 20, 131072, 399999, 64, 250000, 1, 0x60000};
        std::vector<std::vector<uint8_t>> data(sizes.size());
        std::vector<uint8_t> want(sizes.size() * 64);
        uint32_t seed = 99;
        for (size_t k = 0; k < sizes.size(); ++k) {
            data[k].resize(sizes[k]);
            for (auto& b : data[k]) { seed = seed * 1664525u + 1013904223u; b = (uint8_t)(seed >> 24); }
            HostSha hs;
            host_sha512_init(hs);
            host_sha512_update(hs, data[k].data(), data[k].size());
            host_sha512_final(hs, want.data() + 64 * k);
        }
        for (unsigned workers : {1u, 3u, 16u}) {
            MemberHashers mh;
            mh.start(sizes, workers);
            std::vector<uint8_t> slot[2] = {std::vector<uint8_t>(kSlot), std::vector<uint8_t>(kSlot)};
            size_t member = 0;
            uint64_t off = 0; // within the member
            for (unsigned round = 0; member < sizes.size(); ++round) {
                const int b = (int)(round & 1u);
                mh.wait_slot(b);
                size_t at = 0;
                std::vector<MemberHashers::Task> tasks;
                while (member < sizes.size() && at < kSlot) {
                    const uint64_t take = std::min<uint64_t>(kSlot - at, sizes[member] - off);
                    memcpy(slot[b].data() + at, data[member].data() + off, take);
                    tasks.push_back(MemberHashers::Task{(uint32_t)member, slot[b].data() + at, take, off == 0, off + take == sizes[member], b});
                    at += take;
                    off += take;
                    if (off == sizes[member]) { ++member; off = 0; }
                }
                for (const auto& t : tasks) mh.give(t);
            }
            mh.wait_all();
            mh.stop();
            if (mh.digests.size() != want.size() || memcmp(mh.digests.data(), want.data(), want.size()) != 0) return 23;
            uint64_t total = 0;
            for (uint64_t z : sizes) total += z;
            if (mh.bytes != total) return 24;
        }
        { // stopped with work still queued (a pass that failed half-way): what is queued is done, nothing hangs
            MemberHashers mh;
            mh.start(sizes, 2);
            for (size_t k = 0; k < sizes.size(); ++k) mh.give(MemberHashers::Task{(uint32_t)k, data[k].data(), sizes[k], true, true, (int)(k & 1)});
        }
    }
    printf("tsan driver ok\n");
    return 0;
}
