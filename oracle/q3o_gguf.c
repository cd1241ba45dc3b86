/*
 * q3o_gguf.c -- ORACLE (test infrastructure): GGUF container reader.
 * Follows the reference's mini-reader (/root/reference/src/assets_manager.rs:33-148: magic, version>=2,
 * u64 counts, KV value types 0-8/10-12, tensor infos, 32-byte data alignment) and extends it with
 * array-typed KVs (type 9) per the public GGUF spec [EXT], which llama.cpp-format model files need.
 */
#define _GNU_SOURCE
#include "q3o.h"
#include <stdio.h>
#include <stdlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>

typedef struct { const uint8_t* p; const uint8_t* end; int bad; } rd_t;

static uint64_t rd_u(rd_t* r, int n) {
    if (r->p + n > r->end) { r->bad = 1; return 0; }
    uint64_t v = 0;
    memcpy(&v, r->p, (size_t)n);
    r->p += n;
    return v;
}
static int rd_str(rd_t* r, char* out, size_t cap, char** heap) {
    uint64_t len = rd_u(r, 8);
    if (r->bad || r->p + len > r->end) { r->bad = 1; return -1; }
    if (out) {
        size_t c = len < cap - 1 ? (size_t)len : cap - 1;
        memcpy(out, r->p, c);
        out[c] = 0;
    }
    if (heap) {
        *heap = (char*)malloc(len + 1);
        memcpy(*heap, r->p, len);
        (*heap)[len] = 0;
    }
    r->p += len;
    return 0;
}
static int scalar_size(int t) {
    switch (t) {
        case 0: case 1: case 7: return 1;
        case 2: case 3: return 2;
        case 4: case 5: case 6: return 4;
        case 10: case 11: case 12: return 8;
        default: return -1;
    }
}

size_t q3o_type_row_bytes(int type, int64_t k) {
    switch (type) {
        case Q3_T_F32: return (size_t)k * 4;
        case Q3_T_F16: case Q3_T_BF16: return (size_t)k * 2;
        case Q3_T_Q8_0: return (size_t)(k / 32) * 34;
        case Q3_T_Q5_K: return (size_t)(k / 256) * 176;
        case Q3_T_Q6_K: return (size_t)(k / 256) * 210;
        default: return 0;
    }
}

static void read_scalar(rd_t* r, int t, q3o_gguf_kv* kv) {
    uint64_t raw = rd_u(r, scalar_size(t));
    switch (t) {
        case 0: case 2: case 4: case 10: case 7: kv->v.u = raw; break;
        case 1: kv->v.i = (int8_t)raw; break;
        case 3: kv->v.i = (int16_t)raw; break;
        case 5: kv->v.i = (int32_t)raw; break;
        case 11: kv->v.i = (int64_t)raw; break;
        case 6: { float f; uint32_t u = (uint32_t)raw; memcpy(&f, &u, 4); kv->v.f = f; break; }
        case 12: { double d; memcpy(&d, &raw, 8); kv->v.f = d; break; }
    }
}

q3o_gguf* q3o_gguf_open(const char* path, char* err, size_t errlen) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { snprintf(err, errlen, "open %s failed", path); return NULL; }
    struct stat st;
    fstat(fd, &st);
    uint8_t* map = (uint8_t*)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { close(fd); snprintf(err, errlen, "mmap %s failed", path); return NULL; }
    q3o_gguf* g = (q3o_gguf*)calloc(1, sizeof(*g));
    g->map = map; g->map_size = (size_t)st.st_size; g->fd = fd;
    rd_t r = { map, map + st.st_size, 0 };
    if (st.st_size < 24 || memcmp(map, "GGUF", 4) != 0) { snprintf(err, errlen, "Not a GGUF file"); goto fail; }
    r.p += 4;
    g->version = (uint32_t)rd_u(&r, 4);
    if (g->version < 2) { snprintf(err, errlen, "Unsupported GGUF version: %u", g->version); goto fail; }
    g->n_tensors = rd_u(&r, 8);
    g->n_kv = rd_u(&r, 8);
    if (g->n_tensors > (1u << 20) || g->n_kv > (1u << 20)) { snprintf(err, errlen, "implausible counts"); goto fail; }
    g->kv = (q3o_gguf_kv*)calloc(g->n_kv ? g->n_kv : 1, sizeof(q3o_gguf_kv));
    g->tensors = (q3o_gguf_tensor*)calloc(g->n_tensors ? g->n_tensors : 1, sizeof(q3o_gguf_tensor));
    uint64_t alignment = 32;
    for (uint64_t i = 0; i < g->n_kv; i++) {
        q3o_gguf_kv* kv = &g->kv[i];
        rd_str(&r, kv->key, sizeof(kv->key), NULL);
        kv->type = (int)rd_u(&r, 4);
        if (r.bad) break;
        if (kv->type == 8) {
            rd_str(&r, NULL, 0, &kv->str);
        } else if (kv->type == 9) {
            kv->arr_type = (int)rd_u(&r, 4);
            kv->arr_n = rd_u(&r, 8);
            if (kv->arr_type == 8) {
                for (uint64_t j = 0; j < kv->arr_n && !r.bad; j++) rd_str(&r, NULL, 0, NULL);
            } else {
                int sz = scalar_size(kv->arr_type);
                if (sz < 0) { snprintf(err, errlen, "Unknown GGUF array type: %d", kv->arr_type); goto fail; }
                size_t nb = (size_t)sz * kv->arr_n;
                if (r.p + nb > r.end) { r.bad = 1; break; }
                kv->arr = malloc(nb ? nb : 1);
                memcpy(kv->arr, r.p, nb);
                r.p += nb;
            }
        } else if (scalar_size(kv->type) > 0) {
            read_scalar(&r, kv->type, kv);
        } else {
            snprintf(err, errlen, "Unknown GGUF value type: %d", kv->type);
            goto fail;
        }
        if (strcmp(kv->key, "general.alignment") == 0 && kv->type == 4) alignment = kv->v.u;
    }
    for (uint64_t i = 0; i < g->n_tensors && !r.bad; i++) {
        q3o_gguf_tensor* t = &g->tensors[i];
        rd_str(&r, t->name, sizeof(t->name), NULL);
        t->n_dims = (int)rd_u(&r, 4);
        if (t->n_dims > 4) { snprintf(err, errlen, "tensor %s: n_dims %d", t->name, t->n_dims); goto fail; }
        for (int d = 0; d < 4; d++) t->ne[d] = 1;
        for (int d = 0; d < t->n_dims; d++) t->ne[d] = (int64_t)rd_u(&r, 8);
        t->type = (int)rd_u(&r, 4);
        t->offset = rd_u(&r, 8);
    }
    if (r.bad) { snprintf(err, errlen, "truncated GGUF header"); goto fail; }
    {
        size_t pos = (size_t)(r.p - map);
        size_t pad = (alignment - (pos % alignment)) % alignment;
        g->data_start = pos + pad;
    }
    for (uint64_t i = 0; i < g->n_tensors; i++) {
        q3o_gguf_tensor* t = &g->tensors[i];
        int64_t rows = t->ne[1] * t->ne[2] * t->ne[3];
        size_t rb = q3o_type_row_bytes(t->type, t->ne[0]);
        if (rb == 0) { snprintf(err, errlen, "Unsupported tensor type: %d (%s)", t->type, t->name); goto fail; }
        t->nbytes = rb * (size_t)rows;
        if (g->data_start + t->offset + t->nbytes > g->map_size) { snprintf(err, errlen, "tensor %s out of file", t->name); goto fail; }
        t->data = map + g->data_start + t->offset;
    }
    return g;
fail:
    q3o_gguf_close(g);
    return NULL;
}

void q3o_gguf_close(q3o_gguf* g) {
    if (!g) return;
    if (g->kv) {
        for (uint64_t i = 0; i < g->n_kv; i++) { free(g->kv[i].str); free(g->kv[i].arr); }
        free(g->kv);
    }
    free(g->tensors);
    if (g->map) munmap(g->map, g->map_size);
    if (g->fd >= 0) close(g->fd);
    free(g);
}

const q3o_gguf_tensor* q3o_gguf_find(const q3o_gguf* g, const char* name) {
    for (uint64_t i = 0; i < g->n_tensors; i++)
        if (strcmp(g->tensors[i].name, name) == 0) return &g->tensors[i];
    return NULL;
}
const q3o_gguf_kv* q3o_gguf_kv_find(const q3o_gguf* g, const char* key) {
    for (uint64_t i = 0; i < g->n_kv; i++)
        if (strcmp(g->kv[i].key, key) == 0) return &g->kv[i];
    return NULL;
}
