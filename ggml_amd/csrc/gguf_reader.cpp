// gguf_reader.cpp — GGUF metadata parser over a read-only mmap; the C-ABI of include/ggml_cdna4_gguf.h.
//
// Restates the reader of the reference (src/gguf.cpp:319-705: header, key/value pairs, tensor infos, data-section layout) with
// the same acceptance checks, but without stdio reads or payload copies: the file is mapped once and parsed in place with
// a bounds-checked cursor; tensor payloads stay in the mapping, from where they go straight to the HBM weight buffers.
// Host code only (no device code in this translation unit).
#include "../../include/ggml_cdna4_gguf.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

int cdna4_set_error_msg(const char *msg);          // capi.hip (thread-local message behind ggml_cdna4_last_error())

namespace {

constexpr uint32_t kVersionMax = 3;                // GGUF_VERSION, include/gguf.h:42
constexpr size_t   kDefaultAlignment = 32;         // GGUF_DEFAULT_ALIGNMENT, include/gguf.h:46
constexpr size_t   kMaxName = 64;                  // GGML_MAX_NAME, include/ggml.h:227
constexpr int      kMaxDims = 4;                   // GGML_MAX_DIMS
constexpr int      kTypeCount = 39;                // GGML_TYPE_COUNT, include/ggml.h:390

// {elements per block, bytes per block} of enum ggml_type 0..38 (type_traits[], src/ggml.c:565-850); {0,0} = removed type.
// tests/test_gguf.py checks this table against ggml_blck_size / ggml_type_size of the compiled reference.
const struct { int64_t blck; size_t size; } kTypes[kTypeCount] = {
    {1, 4}, {1, 2}, {32, 18}, {32, 20}, {0, 0}, {0, 0}, {32, 22}, {32, 24}, {32, 34}, {32, 36},          // f32 f16 q4_0 q4_1 - - q5_0 q5_1 q8_0 q8_1
    {256, 84}, {256, 110}, {256, 144}, {256, 176}, {256, 210}, {256, 292},                                // q2_K q3_K q4_K q5_K q6_K q8_K
    {256, 66}, {256, 74}, {256, 98}, {256, 50}, {32, 18}, {256, 110}, {256, 82}, {256, 136},              // iq2_xxs iq2_xs iq3_xxs iq1_s iq4_nl iq3_s iq2_s iq4_xs
    {1, 1}, {1, 2}, {1, 4}, {1, 8}, {1, 8}, {256, 56}, {1, 2},                                            // i8 i16 i32 i64 f64 iq1_m bf16
    {0, 0}, {0, 0}, {0, 0}, {256, 54}, {256, 66}, {0, 0}, {0, 0}, {0, 0},                                 // - - - tq1_0 tq2_0 - - -
};

// bytes of a fixed-size value type (gguf_type_size, src/gguf.cpp:69-87); 0 for STRING / ARRAY / out of range
size_t value_size(int t) {
    switch (t) {
        case GGML_CDNA4_GGUF_UINT8: case GGML_CDNA4_GGUF_INT8: case GGML_CDNA4_GGUF_BOOL: return 1;
        case GGML_CDNA4_GGUF_UINT16: case GGML_CDNA4_GGUF_INT16: return 2;
        case GGML_CDNA4_GGUF_UINT32: case GGML_CDNA4_GGUF_INT32: case GGML_CDNA4_GGUF_FLOAT32: return 4;
        case GGML_CDNA4_GGUF_UINT64: case GGML_CDNA4_GGUF_INT64: case GGML_CDNA4_GGUF_FLOAT64: return 8;
        default: return 0;
    }
}

struct KV {
    std::string key;
    bool is_array = false;
    int type = -1;                                 // element type for arrays
    std::vector<uint8_t> data;                     // packed fixed-size elements (bools normalised to 0 / 1)
    std::vector<std::string> strs;                 // string value(s)
    size_t n() const { return type == GGML_CDNA4_GGUF_STRING ? strs.size() : data.size() / value_size(type); }
};

struct TensorInfo {
    std::string name;
    int type = 0;
    int64_t ne[kMaxDims] = {1, 1, 1, 1};
    uint64_t offset = 0;                           // within the data section
    size_t nbytes = 0;
};

int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(const char *fmt, ...) {
    char buf[480];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    return cdna4_set_error_msg(buf);
}

// bounds-checked little-endian cursor over the mapping
struct Cursor {
    const uint8_t *base; size_t size, pos = 0;
    bool take(void *dst, size_t n) {
        if (n > size - pos) return false;
        memcpy(dst, base + pos, n); pos += n; return true;
    }
    template <typename T> bool get(T &v) { return take(&v, sizeof v); }
    bool str(std::string &s) {                     // uint64 length + bytes, no terminator (include/gguf.h:26)
        uint64_t n;
        if (!get(n) || n > size - pos) return false;
        s.assign(reinterpret_cast<const char *>(base + pos), (size_t)n); pos += (size_t)n; return true;
    }
};

inline size_t pad_to(size_t x, size_t a) { return (x + a - 1) & ~(a - 1); }

}  // namespace

struct ggml_cdna4_gguf {
    int fd = -1;
    const uint8_t *map = nullptr; size_t file_size = 0;
    uint32_t version = 0;
    size_t alignment = kDefaultAlignment, data_offset = 0, data_size = 0;
    std::vector<KV> kv;
    std::vector<TensorInfo> info;
    // name -> index: the duplicate checks and the find_* getters are O(1) (the reference scans linearly, which makes a header
    // with millions of keys quadratic)
    std::unordered_map<std::string, int64_t> kv_index, tensor_index;
    ~ggml_cdna4_gguf() {
        if (map) munmap(const_cast<uint8_t *>(map), file_size);
        if (fd >= 0) close(fd);
    }
};

namespace {

int64_t find_key(const ggml_cdna4_gguf *g, const std::string &key) {
    const auto it = g->kv_index.find(key);
    return it == g->kv_index.end() ? -1 : it->second;
}

// one key/value pair at the cursor (src/gguf.cpp:400-459)
bool read_kv(Cursor &c, ggml_cdna4_gguf *g, int64_t i) {
    KV kv;
    if (!c.str(kv.key)) return fail("gguf: file ends inside key %" PRId64, i), false;
    if (kv.key.empty()) return fail("gguf: key %" PRId64 " is empty", i), false;
    if (find_key(g, kv.key) >= 0) return fail("gguf: duplicate key '%s'", kv.key.c_str()), false;
    int32_t type; uint64_t n = 1;
    if (!c.get(type)) return fail("gguf: file ends inside the value type of key '%s'", kv.key.c_str()), false;
    if (type == GGML_CDNA4_GGUF_ARRAY) {
        kv.is_array = true;
        if (!c.get(type) || !c.get(n)) return fail("gguf: file ends inside the array header of key '%s'", kv.key.c_str()), false;
    }
    kv.type = type;
    if (type == GGML_CDNA4_GGUF_STRING) {
        // every string needs at least its 8-byte length: a bound on n that a truncated or hostile file cannot exceed
        if (n > (c.size - c.pos) / 8) return fail("gguf: file ends inside the string array of key '%s'", kv.key.c_str()), false;
        kv.strs.resize((size_t)n);
        for (auto &s : kv.strs) if (!c.str(s)) return fail("gguf: file ends inside a string of key '%s'", kv.key.c_str()), false;
    } else {
        const size_t vs = value_size(type);
        if (vs == 0) return fail("gguf: key '%s' has invalid GGUF type %d", kv.key.c_str(), type), false;
        if (n > (c.size - c.pos) / vs) return fail("gguf: file ends inside the value of key '%s'", kv.key.c_str()), false;
        kv.data.resize((size_t)n * vs);
        c.take(kv.data.data(), kv.data.size());
        if (type == GGML_CDNA4_GGUF_BOOL) for (auto &b : kv.data) b = b != 0;      // gguf_reader::read(bool &), src/gguf.cpp:232-239
    }
    g->kv_index.emplace(kv.key, (int64_t)g->kv.size());
    g->kv.push_back(std::move(kv));
    return true;
}

// one tensor info at the cursor (src/gguf.cpp:478-583)
bool read_tensor_info(Cursor &c, ggml_cdna4_gguf *g, int64_t i) {
    TensorInfo t;
    if (!c.str(t.name)) return fail("gguf: file ends inside tensor name %" PRId64, i), false;
    if (t.name.size() >= kMaxName) return fail("gguf: tensor name %" PRId64 " is too long: %zu >= %zu", i, t.name.size(), kMaxName), false;
    // the reference keeps names in a char[64] and compares them as C strings: an embedded NUL ends the name
    t.name.resize(strlen(t.name.c_str()));
    if (g->tensor_index.count(t.name)) return fail("gguf: duplicate tensor name '%s'", t.name.c_str()), false;
    uint32_t n_dims;
    if (!c.get(n_dims)) return fail("gguf: file ends inside the shape of tensor '%s'", t.name.c_str()), false;
    if (n_dims > (uint32_t)kMaxDims) return fail("gguf: tensor '%s' has invalid number of dimensions: %u > %d", t.name.c_str(), n_dims, kMaxDims), false;
    for (uint32_t j = 0; j < n_dims; j++) {
        if (!c.get(t.ne[j])) return fail("gguf: file ends inside the shape of tensor '%s'", t.name.c_str()), false;
        if (t.ne[j] < 0) return fail("gguf: tensor '%s' dimension %u has invalid number of elements: %" PRId64 " < 0", t.name.c_str(), j, t.ne[j]), false;
    }
    // the total number of elements must be representable (src/gguf.cpp:529-539); a zero-sized dimension divides by zero in
    // the reference, so it is rejected here as well rather than reproduced
    for (int j = 1; j < kMaxDims; j++) if (t.ne[j] == 0) return fail("gguf: tensor '%s' has a zero-sized dimension %d", t.name.c_str(), j), false;
    if (INT64_MAX / t.ne[1] <= t.ne[0] || INT64_MAX / t.ne[2] <= t.ne[0] * t.ne[1] || INT64_MAX / t.ne[3] <= t.ne[0] * t.ne[1] * t.ne[2])
        return fail("gguf: total number of elements in tensor '%s' is not representable", t.name.c_str()), false;
    int32_t type;
    if (!c.get(type)) return fail("gguf: file ends inside the type of tensor '%s'", t.name.c_str()), false;
    if (type < 0 || type >= kTypeCount) return fail("gguf: tensor '%s' has invalid ggml type %d", t.name.c_str(), type), false;
    t.type = type;
    const int64_t blck = kTypes[type].blck; const size_t tsz = kTypes[type].size;
    if (blck == 0 || t.ne[0] % blck != 0)
        return fail("gguf: tensor '%s' of type %d has %" PRId64 " elements per row, not a multiple of block size (%" PRId64 ")", t.name.c_str(), type, t.ne[0], blck), false;
    // ggml_nbytes of the contiguous tensor (src/ggml.c:1153-1170 with the nb[] of src/gguf.cpp:565-569)
    const size_t row = tsz * (size_t)(t.ne[0] / blck);
    const size_t nb[kMaxDims] = {tsz, row, row * (size_t)t.ne[1], row * (size_t)t.ne[1] * (size_t)t.ne[2]};
    if (blck == 1) { t.nbytes = tsz; for (int j = 0; j < kMaxDims; j++) t.nbytes += (size_t)(t.ne[j] - 1) * nb[j]; }
    else { t.nbytes = (size_t)t.ne[0] * nb[0] / (size_t)blck; for (int j = 1; j < kMaxDims; j++) t.nbytes += (size_t)(t.ne[j] - 1) * nb[j]; }
    if (!c.get(t.offset)) return fail("gguf: file ends inside the offset of tensor '%s'", t.name.c_str()), false;
    g->tensor_index.emplace(t.name, (int64_t)g->info.size());
    g->info.push_back(std::move(t));
    return true;
}

bool parse(ggml_cdna4_gguf *g, int require_data) {
    Cursor c{g->map, g->file_size};
    char magic[4];
    if (!c.take(magic, 4)) return fail("gguf: failed to read magic"), false;
    if (memcmp(magic, "GGUF", 4) != 0) return fail("gguf: invalid magic characters, expected 'GGUF'"), false;
    int64_t n_tensors = 0, n_kv = 0;
    if (!c.get(g->version)) return fail("gguf: failed to read header"), false;
    if (g->version == 1) return fail("gguf: GGUFv1 is no longer supported"), false;
    if (g->version > kVersionMax) return fail("gguf: file is version %u but only versions up to %u are supported", g->version, kVersionMax), false;
    if (!c.get(n_tensors) || !c.get(n_kv)) return fail("gguf: failed to read header"), false;
    // every tensor info takes at least 8+4+4+8 bytes and every pair at least 8+4+1: counts beyond that cannot be honest
    if (n_tensors < 0 || (uint64_t)n_tensors > g->file_size / 24) return fail("gguf: number of tensors is %" PRId64 ", more than the file can hold", n_tensors), false;
    if (n_kv < 0 || (uint64_t)n_kv > g->file_size / 13) return fail("gguf: number of key value pairs is %" PRId64 ", more than the file can hold", n_kv), false;
    g->kv.reserve((size_t)n_kv); g->info.reserve((size_t)n_tensors);
    for (int64_t i = 0; i < n_kv; i++) if (!read_kv(c, g, i)) return false;

    const int64_t ai = find_key(g, "general.alignment");         // GGUF_KEY_GENERAL_ALIGNMENT, src/gguf.cpp:464-471
    if (ai >= 0) {
        const KV &a = g->kv[(size_t)ai];
        if (a.is_array || a.type != GGML_CDNA4_GGUF_UINT32) return fail("gguf: general.alignment must be a uint32"), false;   // (GGML_ASSERT in the reference)
        uint32_t v; memcpy(&v, a.data.data(), 4); g->alignment = v;
    }
    if (g->alignment == 0 || (g->alignment & (g->alignment - 1)) != 0) return fail("gguf: alignment %zu is not a power of 2", g->alignment), false;

    for (int64_t i = 0; i < n_tensors; i++) if (!read_tensor_info(c, g, i)) return false;

    // the data section starts at the next multiple of the alignment; every tensor sits at the running padded sum
    g->data_offset = pad_to(c.pos, g->alignment);
    g->data_size = 0;
    for (const auto &t : g->info) {
        if (t.offset != g->data_size) return fail("gguf: tensor '%s' has offset %" PRIu64 ", expected %zu", t.name.c_str(), t.offset, g->data_size), false;
        g->data_size += pad_to(t.nbytes, g->alignment);
    }
    // (an empty data section may start past the end of the file: the reference seeks there and reads nothing)
    if (require_data && g->data_size > 0 && (g->data_offset > g->file_size || g->data_size > g->file_size - g->data_offset))
        return fail("gguf: failed to read tensor data binary blob (file holds %zu of %zu bytes)", g->file_size > g->data_offset ? g->file_size - g->data_offset : (size_t)0, g->data_size), false;
    return true;
}

const KV *kv_at(const ggml_cdna4_gguf *g, int64_t id) {
    if (!g || id < 0 || id >= (int64_t)g->kv.size()) { fail("gguf: key id %" PRId64 " out of range", id); return nullptr; }
    return &g->kv[(size_t)id];
}
const TensorInfo *ti_at(const ggml_cdna4_gguf *g, int64_t id) {
    if (!g || id < 0 || id >= (int64_t)g->info.size()) { fail("gguf: tensor id %" PRId64 " out of range", id); return nullptr; }
    return &g->info[(size_t)id];
}

}  // namespace

extern "C" {

ggml_cdna4_gguf *ggml_cdna4_gguf_open(const char *path, int require_data) {
    if (!path) { fail("gguf: no file name"); return nullptr; }
    ggml_cdna4_gguf *g = new (std::nothrow) ggml_cdna4_gguf;
    if (!g) { fail("gguf: out of memory"); return nullptr; }
    g->fd = open(path, O_RDONLY | O_CLOEXEC);
    struct stat st;
    if (g->fd < 0 || fstat(g->fd, &st) != 0) { fail("gguf: failed to open GGUF file '%s'", path); delete g; return nullptr; }
    g->file_size = (size_t)st.st_size;
    if (g->file_size == 0) { fail("gguf: failed to read magic"); delete g; return nullptr; }
    void *m = mmap(nullptr, g->file_size, PROT_READ, MAP_PRIVATE, g->fd, 0);
    if (m == MAP_FAILED) { fail("gguf: cannot map '%s'", path); delete g; return nullptr; }
    g->map = static_cast<const uint8_t *>(m);
    bool ok = false;
    try { ok = parse(g, require_data); } catch (const std::bad_alloc &) { fail("gguf: out of memory while reading the metadata"); }
    if (!ok) { delete g; return nullptr; }
    // the payloads are about to be streamed front to back into HBM
    madvise(const_cast<uint8_t *>(g->map), g->file_size, MADV_SEQUENTIAL);
    return g;
}

void ggml_cdna4_gguf_close(ggml_cdna4_gguf *g) { delete g; }

uint32_t ggml_cdna4_gguf_version(const ggml_cdna4_gguf *g) { return g->version; }
size_t ggml_cdna4_gguf_alignment(const ggml_cdna4_gguf *g) { return g->alignment; }
size_t ggml_cdna4_gguf_data_offset(const ggml_cdna4_gguf *g) { return g->data_offset; }
size_t ggml_cdna4_gguf_data_size(const ggml_cdna4_gguf *g) { return g->data_size; }

int64_t ggml_cdna4_gguf_n_kv(const ggml_cdna4_gguf *g) { return (int64_t)g->kv.size(); }
int64_t ggml_cdna4_gguf_find_key(const ggml_cdna4_gguf *g, const char *key) { return key ? find_key(g, key) : -1; }
const char *ggml_cdna4_gguf_key(const ggml_cdna4_gguf *g, int64_t id) { const KV *k = kv_at(g, id); return k ? k->key.c_str() : nullptr; }
int ggml_cdna4_gguf_kv_type(const ggml_cdna4_gguf *g, int64_t id) { const KV *k = kv_at(g, id); return !k ? -1 : (k->is_array ? (int)GGML_CDNA4_GGUF_ARRAY : k->type); }
int ggml_cdna4_gguf_arr_type(const ggml_cdna4_gguf *g, int64_t id) {
    const KV *k = kv_at(g, id);
    if (!k) return -1;
    if (!k->is_array) return fail("gguf: key '%s' is not an array", k->key.c_str());
    return k->type;
}
size_t ggml_cdna4_gguf_arr_n(const ggml_cdna4_gguf *g, int64_t id) { const KV *k = kv_at(g, id); return k ? k->n() : 0; }

int ggml_cdna4_gguf_val(const ggml_cdna4_gguf *g, int64_t id, int type, void *out) {
    const KV *k = kv_at(g, id);
    if (!k) return -1;
    if (k->is_array || k->type != type || value_size(type) == 0 || !out) return fail("gguf: key '%s' does not hold a scalar of type %d", k->key.c_str(), type);
    memcpy(out, k->data.data(), value_size(type));
    return 0;
}
const char *ggml_cdna4_gguf_val_str(const ggml_cdna4_gguf *g, int64_t id) {
    const KV *k = kv_at(g, id);
    if (!k) return nullptr;
    if (k->is_array || k->type != GGML_CDNA4_GGUF_STRING) { fail("gguf: key '%s' does not hold a string", k->key.c_str()); return nullptr; }
    return k->strs[0].c_str();
}
const void *ggml_cdna4_gguf_arr_data(const ggml_cdna4_gguf *g, int64_t id) {
    const KV *k = kv_at(g, id);
    if (!k) return nullptr;
    if (!k->is_array || k->type == GGML_CDNA4_GGUF_STRING) { fail("gguf: key '%s' is not an array of fixed-size values", k->key.c_str()); return nullptr; }
    return k->data.data();
}
const char *ggml_cdna4_gguf_arr_str(const ggml_cdna4_gguf *g, int64_t id, size_t i) {
    const KV *k = kv_at(g, id);
    if (!k) return nullptr;
    if (!k->is_array || k->type != GGML_CDNA4_GGUF_STRING || i >= k->strs.size()) { fail("gguf: key '%s' has no string element %zu", k->key.c_str(), i); return nullptr; }
    return k->strs[i].c_str();
}

int64_t ggml_cdna4_gguf_n_tensors(const ggml_cdna4_gguf *g) { return (int64_t)g->info.size(); }
int64_t ggml_cdna4_gguf_find_tensor(const ggml_cdna4_gguf *g, const char *name) {
    if (!name) return -1;
    const auto it = g->tensor_index.find(name);
    return it == g->tensor_index.end() ? -1 : it->second;
}
const char *ggml_cdna4_gguf_tensor_name(const ggml_cdna4_gguf *g, int64_t id) { const TensorInfo *t = ti_at(g, id); return t ? t->name.c_str() : nullptr; }
int ggml_cdna4_gguf_tensor_type(const ggml_cdna4_gguf *g, int64_t id) { const TensorInfo *t = ti_at(g, id); return t ? t->type : -1; }
int ggml_cdna4_gguf_tensor_ne(const ggml_cdna4_gguf *g, int64_t id, int64_t ne[4]) {
    const TensorInfo *t = ti_at(g, id);
    if (!t || !ne) return -1;
    for (int j = 0; j < kMaxDims; j++) ne[j] = t->ne[j];
    return 0;
}
size_t ggml_cdna4_gguf_tensor_offset(const ggml_cdna4_gguf *g, int64_t id) { const TensorInfo *t = ti_at(g, id); return t ? (size_t)t->offset : 0; }
size_t ggml_cdna4_gguf_tensor_size(const ggml_cdna4_gguf *g, int64_t id) { const TensorInfo *t = ti_at(g, id); return t ? t->nbytes : 0; }
const void *ggml_cdna4_gguf_tensor_data(const ggml_cdna4_gguf *g, int64_t id) {
    const TensorInfo *t = ti_at(g, id);
    if (!t) return nullptr;
    const size_t begin = g->data_offset + (size_t)t->offset;
    if (begin > g->file_size || t->nbytes > g->file_size - begin) { fail("gguf: the file ends before the data of tensor '%s'", t->name.c_str()); return nullptr; }
    return g->map + begin;
}

int64_t ggml_cdna4_gguf_blck_size(int type) { return type >= 0 && type < kTypeCount ? kTypes[type].blck : 0; }
size_t ggml_cdna4_gguf_type_size(int type) { return type >= 0 && type < kTypeCount ? kTypes[type].size : 0; }

}  // extern "C"
