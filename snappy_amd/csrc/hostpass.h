// hostpass.h -- internal C++ interface of the host-side pass (walk, YAML, LPT).
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/snaphash.h"

namespace snaphash {

struct Record {          // one fileHash (snappy/hashes.go:93-101) before hashing
    std::string name;    // relative to the walk root
    std::string path;    // root-joined
    uint32_t st_mode = 0;
    bool is_regular = false;
    int64_t size = 0;
    std::string sha512_hex; // only for records parsed from a hashes.yaml
};

struct ParsedRecord {
    std::string name, sha512_hex;
    int64_t size = 0;
    uint32_t st_mode = 0;
    bool has_name = false, has_size = false, has_mode = false;
};
struct ParsedHashes {
    bool has_archive = false;
    std::string archive_hex;
    std::vector<ParsedRecord> files;
};

int mode_string(uint32_t st_mode, char out[11]);
int mode_parse(const char* s, uint32_t* st_mode);
int walk_tree(const char* build_dir, std::vector<Record>& out, int* err_no);
struct WalkEntry;
int records_from_entries(const std::vector<WalkEntry>& ents, std::vector<Record>& out);
bool plain_safe_name(const std::string& s);
// yamlscalar.cpp: the value of `name:` as yaml.v2 writes it (plain, single- or double-quoted, folded at 80 columns)
int yaml_append_name_scalar(const std::string& s, int column, int indent, std::string& out);
bool name_emittable(const std::string& s);
size_t first_unemittable_name(const std::vector<Record>& recs); // recs.size() = every name can be written
void hex_lower(const uint8_t d[64], char out[128]);
int emit_yaml(const std::vector<Record>& recs, const uint8_t archive_digest[64], const uint8_t* file_digests,
              std::string& out);
// The same document written BEFORE the digests exist (round 5: the YAML of a tree no longer waits behind the last kernel):
// every sha512 value is 128 zeros and hex_at says where each one starts -- [0] the archive's, then one per regular record
// in record order.  yaml_fill_digests writes the digests in (ranges on a few threads for a large tree); the result is
// byte for byte what emit_yaml writes.
struct YamlSkeleton {
    std::string text;
    std::vector<size_t> hex_at;
};
int emit_yaml_skeleton(const std::vector<Record>& recs, YamlSkeleton& out, unsigned max_threads = 8 /* 1: on the calling thread alone (a background build beside a pass that needs the cores) */);
void yaml_fill_digests(YamlSkeleton& sk, const uint8_t archive_digest[64], const uint8_t* file_digests);
// ---- the ranks of a one-process-per-GPU job SHARE the walk (snaphash.h, ABI 5) ----
// shard_listing: rank r's share of the walk -- the root listed, the subtrees of the root's entries i with i mod world == r
// walked (walk.h; nothing below an entry whose name begins with "DEBIAN": build.go:229) -- as a self-describing blob.
// records_from_listings: all `world` blobs, in rank order, back into the records of the WHOLE tree in filepath.Walk's
// order.  The blobs come from peer ranks, but every length in them is checked: SNAPHASH_EPARSE for anything malformed,
// SNAPHASH_EMISMATCH when the ranks listed different roots, SNAPHASH_EMODE as the serial loop would raise it.
int shard_listing(const char* build_dir, uint32_t rank, uint32_t world, std::string& blob, int* err_no);
int records_from_listings(const char* build_dir, uint32_t world, const void* const* blobs, const size_t* blob_lens, std::vector<Record>& recs);
int parse_yaml(const char* text, size_t len, ParsedHashes& out);
bool digest_matches_hex(const uint8_t d[64], const std::string& hex);
int lpt_assign(const uint64_t* lens, size_t n, int nshards, int32_t* shard_of);

} // namespace snaphash
