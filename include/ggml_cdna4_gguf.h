/*
 * ggml_cdna4_gguf.h — C-ABI of the GGUF reader in `libcdna4_kernels.so` (the on-disk format next to the
 * MUL_MAT path — the tensor payloads of a GGUF file ARE the block arrays ggml_cdna4_mul_mat consumes).
 *
 * What it replaces in the reference (paths relative to the reference tree):
 *   ggml_cdna4_gguf_open            gguf_init_from_file                      src/gguf.cpp:319-705, include/gguf.h:80
 *   ggml_cdna4_gguf_close           gguf_free                                src/gguf.cpp:707-712
 *   ..._version/_alignment/_data_offset                                      include/gguf.h:87-89
 *   ..._n_kv/_find_key/_key/_kv_type/_arr_type/_arr_n                        include/gguf.h:91-96,112
 *   ..._val (one entry point for the eleven fixed-size scalar types)         gguf_get_val_u8 .. _bool, include/gguf.h:99-109
 *   ..._val_str/_arr_data/_arr_str                                           include/gguf.h:110,116,119
 *   ..._n_tensors/_find_tensor/_tensor_name/_tensor_type/_tensor_offset/_tensor_size   include/gguf.h:121-126
 *   ..._tensor_ne                   the ne[4] the reader fills per tensor    src/gguf.cpp:505-539
 *   ..._tensor_data                 `cur->data = data->data + info.offset`   src/gguf.cpp:676-679
 *
 * Design: the file is mmap'ed read-only and parsed in place with a bounds-checked cursor; only the metadata (keys, strings,
 * small arrays, tensor table) is copied.  Tensor payloads are never copied on the host: ggml_cdna4_gguf_tensor_data() is a
 * pointer into the mapping, ready for hipMemcpyAsync / ggml_backend_tensor_set into the HBM-resident weight buffer.
 *
 * Validation is the reference's, check for check (magic, version 2..3, counts, duplicate keys / tensor names, value types,
 * name length < 64, n_dims <= 4, ne >= 0, element-count overflow, tensor type range, row % block size, alignment a power of two,
 * tensor offsets = the running padded sum).  A file the reference rejects is rejected here (tests/test_gguf.py runs both on
 * the same corrupted files).  Differences, all on malformed input or programmer error: this library never aborts — an empty
 * key (GGML_ASSERT in the reference, src/gguf.cpp:131), a value getter called with the wrong type or an index out of range
 * return NULL / -1 and set ggml_cdna4_last_error() instead of GGML_ASSERT'ing.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* value types: numerically identical to enum gguf_type (include/gguf.h:52-67) */
enum ggml_cdna4_gguf_type {
    GGML_CDNA4_GGUF_UINT8 = 0, GGML_CDNA4_GGUF_INT8 = 1, GGML_CDNA4_GGUF_UINT16 = 2, GGML_CDNA4_GGUF_INT16 = 3,
    GGML_CDNA4_GGUF_UINT32 = 4, GGML_CDNA4_GGUF_INT32 = 5, GGML_CDNA4_GGUF_FLOAT32 = 6, GGML_CDNA4_GGUF_BOOL = 7,
    GGML_CDNA4_GGUF_STRING = 8, GGML_CDNA4_GGUF_ARRAY = 9, GGML_CDNA4_GGUF_UINT64 = 10, GGML_CDNA4_GGUF_INT64 = 11,
    GGML_CDNA4_GGUF_FLOAT64 = 12, GGML_CDNA4_GGUF_TYPE_COUNT = 13,
};

typedef struct ggml_cdna4_gguf ggml_cdna4_gguf;

/* require_data != 0: fail if the file is shorter than its tensor data section (gguf_init_from_file with a ggml context and
 * no_alloc = false); 0: metadata only (no_alloc = true) — tensor_data() then returns NULL for payloads beyond the end of file.
 * Returns NULL on any failure; ggml_cdna4_last_error() says why. */
ggml_cdna4_gguf * ggml_cdna4_gguf_open(const char * path, int require_data);
void              ggml_cdna4_gguf_close(ggml_cdna4_gguf * g);

uint32_t ggml_cdna4_gguf_version    (const ggml_cdna4_gguf * g);
size_t   ggml_cdna4_gguf_alignment  (const ggml_cdna4_gguf * g);
size_t   ggml_cdna4_gguf_data_offset(const ggml_cdna4_gguf * g);   /* file offset of the tensor data section */
size_t   ggml_cdna4_gguf_data_size  (const ggml_cdna4_gguf * g);   /* its size: the padded sum of the tensor sizes */

int64_t      ggml_cdna4_gguf_n_kv    (const ggml_cdna4_gguf * g);
int64_t      ggml_cdna4_gguf_find_key(const ggml_cdna4_gguf * g, const char * key);         /* -1 if absent */
const char * ggml_cdna4_gguf_key     (const ggml_cdna4_gguf * g, int64_t key_id);
int          ggml_cdna4_gguf_kv_type (const ggml_cdna4_gguf * g, int64_t key_id);           /* GGUF_ARRAY for arrays */
int          ggml_cdna4_gguf_arr_type(const ggml_cdna4_gguf * g, int64_t key_id);           /* element type of an array */
size_t       ggml_cdna4_gguf_arr_n   (const ggml_cdna4_gguf * g, int64_t key_id);
/* scalar value of exactly `type` (a fixed-size type; bool is one byte, 0 or 1) copied to `out`; 0, or -1 on a type mismatch */
int          ggml_cdna4_gguf_val     (const ggml_cdna4_gguf * g, int64_t key_id, int type, void * out);
const char * ggml_cdna4_gguf_val_str (const ggml_cdna4_gguf * g, int64_t key_id);
const void * ggml_cdna4_gguf_arr_data(const ggml_cdna4_gguf * g, int64_t key_id);           /* non-string arrays, packed elements */
const char * ggml_cdna4_gguf_arr_str (const ggml_cdna4_gguf * g, int64_t key_id, size_t i);

int64_t      ggml_cdna4_gguf_n_tensors    (const ggml_cdna4_gguf * g);
int64_t      ggml_cdna4_gguf_find_tensor  (const ggml_cdna4_gguf * g, const char * name);   /* -1 if absent */
const char * ggml_cdna4_gguf_tensor_name  (const ggml_cdna4_gguf * g, int64_t tensor_id);
int          ggml_cdna4_gguf_tensor_type  (const ggml_cdna4_gguf * g, int64_t tensor_id);   /* enum ggml_type */
int          ggml_cdna4_gguf_tensor_ne    (const ggml_cdna4_gguf * g, int64_t tensor_id, int64_t ne[4]);   /* unused dims are 1 */
size_t       ggml_cdna4_gguf_tensor_offset(const ggml_cdna4_gguf * g, int64_t tensor_id);   /* within the data section */
size_t       ggml_cdna4_gguf_tensor_size  (const ggml_cdna4_gguf * g, int64_t tensor_id);   /* ggml_nbytes */
/* host pointer to the payload inside the read-only mapping (valid until close); NULL if the file ends before it does */
const void * ggml_cdna4_gguf_tensor_data  (const ggml_cdna4_gguf * g, int64_t tensor_id);

/* payload of one tensor -> device memory, through two pinned staging buffers (the CPU fills one while the other is in flight
 * on `stream`, a hipStream_t, NULL = default stream); returns when the last chunk has left the staging buffers.  Replaces the
 * reference's blob read + ggml_backend_tensor_set (src/gguf.cpp:644-660, examples/gpt-2/main-backend.cpp:412-420).
 * 0, or a negative status with ggml_cdna4_last_error().  (Not yet GPU-verified: DESIGN.md 4.5.) */
int ggml_cdna4_gguf_upload(const ggml_cdna4_gguf * g, int64_t tensor_id, void * dst_device, size_t dst_bytes, void * stream);

/* block size / bytes per block of a ggml tensor type as stored in GGUF files (ggml_blck_size / ggml_type_size,
 * src/ggml.c:1176-1182); 0 for removed or unknown types */
int64_t ggml_cdna4_gguf_blck_size(int ggml_type);
size_t  ggml_cdna4_gguf_type_size(int ggml_type);

#ifdef __cplusplus
}
#endif
