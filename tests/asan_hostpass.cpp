// Test-only driver: snappy_amd/csrc/hostpass.cpp under AddressSanitizer + UBSan (CPU build; GPU
// sanitizers are not available on the pool).  Walks a tree, emits and re-parses its YAML, then
// feeds the parser -- which reads untrusted hashes.yaml from packages -- thousands of mutated
// documents.  Exit code 0 = no sanitizer report and all invariants held.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../snappy_amd/csrc/hostpass.h"
#include "../snappy_amd/csrc/planner.h"

using namespace snaphash;

static uint64_t rng_state = 0x9e3779b97f4a7c15ULL;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::vector<Record> recs;
    int en = 0;
    if (walk_tree(argv[1], recs, &en) != SNAPHASH_OK) return 3;
    std::vector<uint8_t> dig(64 * (recs.size() + 1), 0xab);
    std::string y;
    if (emit_yaml(recs, dig.data(), dig.data() + 64, y) != SNAPHASH_OK) return 4;
    ParsedHashes ph;
    if (parse_yaml(y.data(), y.size(), ph) != SNAPHASH_OK || ph.files.size() != recs.size()) return 5;
    for (size_t i = 0; i < recs.size(); ++i)
        if (ph.files[i].name != recs[i].name) return 6;

    FILE* f = fopen(argv[2], "rb");
    if (!f) return 7;
    std::string golden;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) golden.append(buf, n);
    fclose(f);
    static const char* const frag[] = {"\n- name: ", "\n  mode: ", "'", "\"", "\\", "\\x", "\\u12", ": ", "files:", "[]", "{}",
                                       "\n  xattr:\n    a: b", "\n  size: ", "-", "#", "\t", "\r\n", "archive-sha512: ", "  "};
    int ok = 0, bad = 0;
    for (int it = 0; it < 20000; ++it) {
        std::string m = (it & 1) ? golden : y;
        const int edits = 1 + rnd() % 4;
        for (int e = 0; e < edits && !m.empty(); ++e) {
            const size_t pos = rnd() % m.size();
            switch (rnd() % 5) {
            case 0: m[pos] = (char)(rnd() & 0xff); break;
            case 1: m.erase(pos, 1 + rnd() % 8); break;
            case 2: m.insert(pos, frag[rnd() % (sizeof frag / sizeof *frag)]); break;
            case 3: m.resize(pos); break;
            default: m.insert(pos, 1 + rnd() % 3, (char)(rnd() & 0xff)); break;
            }
        }
        ParsedHashes p;
        const int rc = parse_yaml(m.data(), m.size(), p);
        if (rc == SNAPHASH_OK) {
            ++ok;
            for (const ParsedRecord& r : p.files) {
                char ms[11];
                if (!r.has_name || !r.has_mode || mode_string(r.st_mode, ms) != SNAPHASH_OK) return 8; // accepted => well-formed
            }
        } else if (rc == SNAPHASH_EPARSE) {
            ++bad;
        } else {
            return 9;
        }
    }
    // The name emitter (yamlscalar.cpp) against the parser: random names out of an alphabet rich in YAML indicators,
    // spaces (folding), quotes, escapes and multi-byte characters.  Whatever is emitted must parse back to the name.
    static const char* const piece[] = {"a", "b", " ", "  ", ":", ": ", "#", " #", "'", "\"", "\\", "-", "?", "~", "1", "0x", ".", "e", "_",
                                        "\t", "\x01", "\xc3\xa9", "\xe2\x82\xac", "\xf0\x9f\x98\x80", "\xef\xbb\xbf", "\xff", "word ", "@", "!", "&",
                                        "*", "|", ">", "%", "`", "[", "]", "{", "}", ",", "true", "null", "+", "\xc2\xa0", "\x7f"};
    int emitted = 0, refused = 0;
    for (int it = 0; it < 20000; ++it) {
        std::string name;
        const int np = 1 + rnd() % ((it % 7 == 0) ? 60 : 8);
        for (int k = 0; k < np; ++k) name += piece[rnd() % (sizeof piece / sizeof *piece)];
        std::string doc = "archive-sha512: 00\nfiles:\n- name:";
        const int rc = yaml_append_name_scalar(name, 7, 4, doc);
        if (rc == SNAPHASH_ENAME) { ++refused; continue; }
        if (rc != SNAPHASH_OK) return 12;
        ++emitted;
        doc += "\n  mode: frw-r--r--\n";
        ParsedHashes p;
        if (parse_yaml(doc.data(), doc.size(), p) != SNAPHASH_OK || p.files.size() != 1 || p.files[0].name != name) {
            fprintf(stderr, "name round trip failed for [%s]\n%s", name.c_str(), doc.c_str());
            return 13;
        }
    }
    // The shared walk's listings (round 5: snaphash_shard_list / _plan_from, hostpass.cpp shard_listing / records_from_listings):
    // three ranks' blobs of the tree must give the full walk's records, and 20 000 mutated sets must come back as an error
    // or as well-formed records -- never be read past their end (this build is AddressSanitizer's).
    int lists_ok = 0, lists_bad = 0;
    {
        const uint32_t world = 3;
        std::string blob[3];
        for (uint32_t r = 0; r < world; ++r) {
            int e2 = 0;
            if (shard_listing(argv[1], r, world, blob[r], &e2) != SNAPHASH_OK) return 30;
        }
        const void* ptr[3] = {blob[0].data(), blob[1].data(), blob[2].data()};
        size_t len[3] = {blob[0].size(), blob[1].size(), blob[2].size()};
        std::vector<Record> again;
        if (records_from_listings(argv[1], world, ptr, len, again) != SNAPHASH_OK || again.size() != recs.size()) return 31;
        for (size_t i = 0; i < recs.size(); ++i)
            if (again[i].name != recs[i].name || again[i].path != recs[i].path || again[i].st_mode != recs[i].st_mode || again[i].size != recs[i].size ||
                again[i].is_regular != recs[i].is_regular) return 32;
        for (int it = 0; it < 20000; ++it) {
            std::string m[3] = {blob[0], blob[1], blob[2]};
            std::string& v = m[rnd() % 3];
            const int edits = 1 + rnd() % 3;
            for (int e = 0; e < edits && !v.empty(); ++e) {
                const size_t pos = rnd() % v.size();
                switch (rnd() % 5) {
                case 0: v[pos] = (char)(rnd() & 0xff); break;
                case 1: v.erase(pos, 1 + rnd() % 8); break;
                case 2: v.resize(pos); break;
                case 3: { uint32_t big = 0xfffffff0u + (uint32_t)(rnd() % 16); if (pos + 4 <= v.size()) memcpy(&v[pos & ~(size_t)3], &big, 4); break; } // a length field blown up
                default: v.insert(pos, 1 + rnd() % 3, (char)(rnd() & 0xff)); break;
                }
            }
            const void* mp[3] = {m[0].data(), m[1].data(), m[2].data()};
            size_t ml[3] = {m[0].size(), m[1].size(), m[2].size()};
            std::vector<Record> out;
            const int rc = records_from_listings(argv[1], world, mp, ml, out);
            if (rc == SNAPHASH_OK) {
                ++lists_ok;
                for (const Record& r : out) { char ms[11]; if (r.name.empty() || mode_string(r.st_mode, ms) != SNAPHASH_OK) return 33; }
            } else if (rc == SNAPHASH_EPARSE || rc == SNAPHASH_EMISMATCH || rc == SNAPHASH_EMODE) {
                ++lists_bad;
            } else {
                return 34;
            }
        }
    }
    uint32_t mode;
    if (mode_parse("", &mode) == SNAPHASH_OK) return 10;
    std::vector<uint64_t> lens(1000);
    std::vector<int32_t> shard(1000);
    for (auto& l : lens) l = rnd() % (1u << 26);
    if (lpt_assign(lens.data(), lens.size(), 8, shard.data()) != SNAPHASH_OK) return 11;
    // The planner (planner.cpp) on random stream lists and models, degenerate ones included: every stream is planned exactly
    // once, the sums add up, a plan never asks for more threads than the model allows or than it has streams for.
    int plans = 0;
    for (int it = 0; it < 3000; ++it) {
        const size_t n = (it % 50 == 0) ? 0 : 1 + rnd() % ((it % 9 == 0) ? 40000 : 300);
        std::vector<uint64_t> pl(n);
        for (auto& l : pl) {
            switch (rnd() % 6) {
            case 0: l = 0; break;
            case 1: l = rnd() % 300; break;
            case 2: l = (uint64_t)rnd() << (rnd() % 8); break; // up to 2^39
            case 3: l = 1u << 20; break;
            default: l = rnd() % (1u << 22); break;
            }
        }
        PlanModel pm;
        pm.n_devices = 1 + rnd() % 8;
        pm.cpus = (it % 11 == 0) ? 1 : 1 + rnd() % 300;
        pm.fill_threads = rnd() % 14;
        pm.host_threads = (rnd() % 3 == 0) ? 1 + rnd() % 300 : 0;
        pm.from_files = rnd() & 1;
        if (rnd() % 4 == 0) pm.host_rate = 1e6 * (1 + rnd() % 5000);
        if (rnd() % 4 == 0) pm.gpu_link = 1e8 * (1 + rnd() % 1000);
        if (rnd() % 4 == 0) pm.gpu_latency = 1e-6 * (rnd() % 100000);
        const PlanResult r = plan_streams(pl.data(), n, pm);
        if (r.on_host.size() != n) return 14;
        uint64_t hb = 0, hs = 0;
        for (size_t i = 0; i < n; ++i) { if (r.on_host[i] > 1) return 15; if (r.on_host[i]) { hb += pl[i]; ++hs; } }
        if (hb != r.host_bytes || hs != r.host_streams) return 16;
        if ((hs == 0) != (r.host_threads == 0)) return 17;
        if (r.host_threads > hs || r.host_threads > std::max(pm.host_threads, pm.cpus)) return 18;
        if (!(r.gpu_seconds >= 0) || !(r.host_seconds >= 0)) return 19;
        if (hs == n && n && r.gpu_seconds != 0) return 20;
        ++plans;
    }
    printf("asan driver ok: %d plans, %zu records, %d mutated documents accepted, %d rejected; %d random names emitted and read back, %d refused; %d mutated listings accepted, %d rejected\n", plans, recs.size(), ok, bad, emitted, refused, lists_ok, lists_bad);
    return 0;
}
