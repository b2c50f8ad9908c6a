// Page-header walk of a column chunk on the device (ref: page_reader::next_page,
// src/column_chunk_reader.cc:27-50; deserialize_thrift_msg, include/parquet4seastar/thrift_serdes.hh:117-154;
// struct PageHeader and the three page-type headers, doc/parquet/parquet.thrift:507-659).
//
// A chunk is one chain: a header says how long its page is, and the next header starts behind that page.  One
// wavefront takes one chunk and the whole grid takes a batch of chunks (the columns and row groups of a file), which
// is where the parallelism of this step lies.  All 64 lanes run the same scalar parse in step (no divergence); what
// they share is the window: 512 bytes of the chunk in LDS, refilled with one coalesced 8-byte load per lane
// whenever the read position leaves it.  A header is a few dozen bytes unless it carries statistics, so a page costs
// one refill (one trip to HBM: pages are far apart) plus a few hundred scalar instructions.
//
// The Thrift compact protocol is read as the generated reader of the reference reads it: known fields by id, every
// other field skipped by wire type (containers and nested structs with an explicit stack, 32 deep at most), required
// fields checked when a struct ends.
#pragma once
#include "p4s_common.cuh"

namespace p4s {

constexpr int HDR_WIN = 512;
constexpr int HDR_MAX_DEPTH = 32;

struct HdrReader {
    const uint8_t *base;      // first byte of the chunk
    uint64_t size;            // bytes of the chunk
    uint64_t room;            // bytes readable from base before the input allocation ends
    uint64_t pos;             // read position, chunk-relative
    int64_t wlo;              // chunk-relative offset of win[0] (window invalid when wlo < 0 ... see ensure)
    bool wvalid;
    uint64_t cache;              // the next bytes of the chunk (see load8)
    uint32_t cidx, cn;
    int32_t err;
    uint64_t err_aux;
    uint8_t *win;             // HDR_WIN bytes of LDS
    struct Frame { uint64_t rem; int32_t last; uint8_t kind, ta, tb, pad; };
    Frame *frames;            // HDR_MAX_DEPTH + 1 of them in LDS: the stack of skip()

    __device__ __forceinline__ void fail(int32_t code, uint64_t aux = 0) {
        if (!err) { err = code; err_aux = aux; }
        cn = 0; cidx = 0;           // the next u8() sees the error
    }
    // makes byte `pos` available in the window (all lanes call this together)
    __device__ __forceinline__ void refill() {
        __builtin_amdgcn_wave_barrier();
        // 8-byte aligned global address; bytes before the chunk belong to the same allocation or are not read
        const uint32_t mis = (uint32_t)((uintptr_t)(base + pos) & 7u);
        wlo = (int64_t)pos - mis;
        const uint32_t lane = lane_id();
        const int64_t o = wlo + (int64_t)lane * 8;
        uint64_t v = 0;
        if (o >= 0 && (uint64_t)o + 8 <= room) {
            v = P4S_LD(reinterpret_cast<const uint64_t *>(base + o), 80);
        } else {
            for (int i = 0; i < 8; ++i) {
                const int64_t q = o + i;
                if (q >= 0 && (uint64_t)q < room) v |= (uint64_t)P4S_LD(base + q, 81) << (8 * i);
            }
        }
        reinterpret_cast<uint64_t *>(win)[lane] = v;
        __builtin_amdgcn_wave_barrier();
        wvalid = true;
    }
    // Up to eight bytes of the chunk from `pos` on, kept in scalar registers: one LDS read serves eight u8() calls.
    // (Every value read from the window is the same in all lanes: said so to the compiler, so that the parse runs
    // on the scalar unit with uniform branches.)  cidx / cn = next byte / valid bytes of the cache; anything that
    // moves `pos` other than u8() drops the cache.
    __device__ __forceinline__ void load8() {
        if (!wvalid || (uint64_t)((int64_t)pos - wlo) > (uint64_t)(HDR_WIN - 8)) refill();
        uint64_t v;
        __builtin_memcpy(&v, win + (uint32_t)((int64_t)pos - wlo), 8);          // LDS reads need no alignment
        cache = ((uint64_t)first_u32((uint32_t)(v >> 32)) << 32) | first_u32((uint32_t)v);
        const uint64_t left = size - pos;
        cn = left < 8 ? (uint32_t)left : 8u;
        cidx = 0;
    }
    __device__ __forceinline__ uint32_t u8() {
        if (cidx >= cn) {
            if (err) return 0;
            if (pos >= size) { fail(P4S_E_THRIFT_EOF); return 0; }
            load8();
        }
        const uint32_t b = (uint32_t)(cache >> (8 * cidx)) & 0xFFu;
        ++cidx;
        ++pos;
        return b;
    }
    __device__ __forceinline__ void seek(uint64_t to) { pos = to; cn = 0; cidx = 0; }
    __device__ __forceinline__ uint64_t varint() {
        uint64_t v = 0;
        for (int sh = 0; sh < 70; sh += 7) {
            const uint32_t b = u8();
            if (err) return 0;
            v |= (uint64_t)(b & 0x7Fu) << (sh & 63);
            if (!(b & 0x80u)) return v;
        }
        fail(P4S_E_THRIFT_VARINT);
        return 0;
    }
    __device__ __forceinline__ int64_t zigzag() {
        const uint64_t u = varint();
        return (int64_t)((u >> 1) ^ (~(u & 1) + 1));
    }
    __device__ __forceinline__ void skip_bytes(uint64_t n) {
        if (err) return;
        if (size - pos < n) { pos = size; fail(P4S_E_THRIFT_EOF); return; }
        seek(pos + n);
    }
    // next field header of the current struct; false at STOP (or on error)
    __device__ __forceinline__ bool field(int32_t &last, int32_t &id, uint32_t &type) {
        const uint32_t h = u8();
        if (err || h == 0) return false;
        type = h & 0x0Fu;
        const uint32_t delta = h >> 4;
        id = delta ? last + (int32_t)delta : (int32_t)zigzag();
        last = (int32_t)(int16_t)id;
        return !err;
    }
    __device__ __forceinline__ int32_t i32(uint32_t type) {
        if (type != 5 && type != 4 && type != 6) { fail(P4S_E_THRIFT_NOT_INT); return 0; }
        return (int32_t)zigzag();
    }
    // Skips one value of wire type `type`.  Containers and structs are walked with an explicit stack; a value nested
    // more than HDR_MAX_DEPTH containers deep is an error (bool elements of containers are single bytes and exempt).
    __device__ __forceinline__ void skip(uint32_t type0) {
        // frame = a container being walked: kind (0 struct, 1 list / set, 2 map), remaining elements, element types.
        // Containers at depths 0 .. HDR_MAX_DEPTH; the stack lives in LDS (every lane writes the same values).
        Frame *const fr = frames;
        int sp = 0;
        uint32_t type = type0;
        bool elem = false;          // the value is a container element (a bool element takes a byte)
        bool have = true;           // a value of `type` at depth sp is to be skipped
        for (;;) {
            if (err) return;
            if (have) {
                have = false;
                if (elem && (type == 1 || type == 2)) { u8(); continue; }
                if (sp > HDR_MAX_DEPTH) { fail(P4S_E_THRIFT_DEPTH); return; }
                switch (type) {
                case 1: case 2: break;
                case 3: u8(); break;
                case 4: case 5: case 6: varint(); break;
                case 7: skip_bytes(8); break;
                case 8: skip_bytes(varint()); break;
                case 9: case 10: {
                    const uint32_t h = u8();
                    uint64_t n = h >> 4;
                    if (n == 15) n = varint();
                    fr[sp] = Frame{n, 0, 1, (uint8_t)(h & 0x0Fu), 0, 0}; ++sp;
                    break;
                }
                case 11: {
                    const uint64_t n = varint();
                    uint32_t kv = 0;
                    if (n) kv = u8();
                    // 2n values, alternating key and value types
                    fr[sp] = Frame{n > 0x7FFFFFFFFFFFFFFFull ? ~0ull - 1 : 2 * n, 0, 2, (uint8_t)(kv >> 4), (uint8_t)(kv & 0x0Fu), 0}; ++sp;
                    break;
                }
                case 12:
                    fr[sp] = Frame{0, 0, 0, 0, 0, 0}; ++sp;
                    break;
                default:
                    fail(P4S_E_THRIFT_WIRE_TYPE, type);
                    return;
                }
                continue;
            }
            if (sp == 0) return;
            const int t = sp - 1;
            const uint32_t k = first_u32(fr[t].kind);
            if (k == 0) {
                int32_t id, lastv = (int32_t)first_u32((uint32_t)fr[t].last);
                uint32_t ft;
                const bool more = field(lastv, id, ft);
                fr[t].last = lastv;
                if (!more) { --sp; continue; }
                type = ft; elem = false; have = true;
            } else {
                const uint64_t r = ((uint64_t)first_u32((uint32_t)(fr[t].rem >> 32)) << 32) | first_u32((uint32_t)fr[t].rem);
                if (r == 0) { --sp; continue; }
                // map frames count down from 2n: an even remainder means a key comes next
                type = first_u32((k == 2 && (r & 1ull)) ? fr[t].tb : fr[t].ta);
                fr[t].rem = r - 1;
                elem = true; have = true;
            }
        }
    }
};

struct HdrPage {
    int32_t type, uncompressed_page_size, compressed_page_size;
    int32_t num_values, encoding, def_enc, rep_enc, def_len, rep_len, num_nulls, num_rows;
    uint32_t flags;
};

// The three page-type structs: `kind` 0 DataPageHeader, 1 DictionaryPageHeader, 2 DataPageHeaderV2.
__device__ __forceinline__ void hdr_read_sub(HdrReader &rd, int kind, HdrPage &pg) {
    int32_t last = 0, id;
    uint32_t type, seen = 0;
    bool compressed_set = false;
    while (rd.field(last, id, type)) {
        if (id >= 1 && id <= 6) seen |= 1u << id;
        if (kind == 0) {
            switch (id) {
            case 1: pg.num_values = rd.i32(type); break;
            case 2: pg.encoding = rd.i32(type); break;
            case 3: pg.def_enc = rd.i32(type); break;
            case 4: pg.rep_enc = rd.i32(type); break;
            default: rd.skip(type);
            }
        } else if (kind == 1) {
            switch (id) {
            case 1: pg.num_values = rd.i32(type); break;
            case 2: pg.encoding = rd.i32(type); break;
            case 3: if (type == 1) pg.flags |= P4S_PAGE_IS_SORTED; else pg.flags &= ~(uint32_t)P4S_PAGE_IS_SORTED; break;
            default: rd.skip(type);
            }
        } else {
            switch (id) {
            case 1: pg.num_values = rd.i32(type); break;
            case 2: pg.num_nulls = rd.i32(type); break;
            case 3: pg.num_rows = rd.i32(type); break;
            case 4: pg.encoding = rd.i32(type); break;
            case 5: pg.def_len = rd.i32(type); break;
            case 6: pg.rep_len = rd.i32(type); break;
            case 7:
                compressed_set = true;
                if (type == 1) pg.flags |= P4S_PAGE_IS_COMPRESSED; else pg.flags &= ~(uint32_t)P4S_PAGE_IS_COMPRESSED;
                break;
            default: rd.skip(type);
            }
        }
        if (rd.err) return;
    }
    if (rd.err) return;
    const uint32_t required = kind == 0 ? 0x1Eu : (kind == 1 ? 0x06u : 0x7Eu);
    if ((seen & required) != required) { rd.fail(P4S_E_THRIFT_INVALID_DATA); return; }
    if (kind == 2 && !compressed_set) pg.flags |= P4S_PAGE_IS_COMPRESSED;      // parquet.thrift: default true
}

__device__ __forceinline__ void hdr_read_page(HdrReader &rd, HdrPage &pg) {
    pg = HdrPage{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0u};
    int32_t last = 0, id;
    uint32_t type, seen = 0;
    while (rd.field(last, id, type)) {
        if (id >= 1 && id <= 3) seen |= 1u << id;
        switch (id) {
        case 1: pg.type = rd.i32(type); break;
        case 2: pg.uncompressed_page_size = rd.i32(type); break;
        case 3: pg.compressed_page_size = rd.i32(type); break;
        case 4: rd.i32(type); pg.flags |= P4S_PAGE_HAS_CRC; break;
        case 5: if (type != 12) rd.skip(type); else { hdr_read_sub(rd, 0, pg); pg.flags |= P4S_PAGE_HAS_DATA_HEADER; } break;
        case 7: if (type != 12) rd.skip(type); else { hdr_read_sub(rd, 1, pg); pg.flags |= P4S_PAGE_HAS_DICT_HEADER; } break;
        case 8: if (type != 12) rd.skip(type); else { hdr_read_sub(rd, 2, pg); pg.flags |= P4S_PAGE_HAS_DATA_HEADER_V2; } break;
        default: rd.skip(type);
        }
        if (rd.err) return;
    }
    if (rd.err) return;
    if ((seen & 0x0Eu) != 0x0Eu) rd.fail(P4S_E_THRIFT_INVALID_DATA);
}

// little-endian int32 at chunk offset `at`, or -1 when the chunk does not hold it
__device__ __forceinline__ int32_t hdr_prefix_at(HdrReader &rd, uint64_t at, uint64_t page_end) {
    if (at + 4 > page_end || at + 4 > rd.size) return -1;
    uint32_t v = 0;
    const uint64_t keep = rd.pos;
    rd.seek(at);
    for (int i = 0; i < 4; ++i) v |= rd.u8() << (8 * i);
    rd.seek(keep);
    return rd.err ? -1 : (int32_t)v;
}

__global__ __launch_bounds__(64) void k_walk_page_headers(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_chunk_desc *__restrict__ chunks,
        p4s_page_info *__restrict__ pages, p4s_page_status *__restrict__ status) {
    __shared__ __align__(8) uint8_t win[HDR_WIN];
    __shared__ HdrReader::Frame frames[HDR_MAX_DEPTH + 1];
    const uint32_t c = blockIdx.x;
    const p4s_chunk_desc cd = chunks[c];
    const uint32_t lane = lane_id();
    HdrReader rd;
    const uint64_t off = cd.offset < in_bytes ? cd.offset : in_bytes;
    rd.base = d_in + off;
    rd.room = in_bytes - off;
    rd.size = cd.size < rd.room ? cd.size : rd.room;
    rd.pos = 0; rd.wlo = 0; rd.wvalid = false; rd.cache = 0; rd.cidx = 0; rd.cn = 0; rd.err = 0; rd.err_aux = 0; rd.win = win; rd.frames = frames;
    p4s_page_info *out = pages + cd.first_page;
    uint32_t n = 0;
    int32_t code = 0;
    uint64_t aux0 = 0, aux1 = 0, at = 0;
    while (rd.pos < rd.size) {
        at = rd.pos;
        HdrPage pg;
        hdr_read_page(rd, pg);
        if (rd.err) {
            code = rd.err;
            aux0 = rd.err == P4S_E_THRIFT_EOF ? rd.size - at : rd.err_aux;
            break;
        }
        if (pg.compressed_page_size < 0) {                                        // column_chunk_reader.cc:33-36
            code = P4S_E_PAGE_NEGATIVE_SIZE; aux0 = (uint64_t)(int64_t)pg.compressed_page_size;
            break;
        }
        const uint64_t body = rd.pos, sz = (uint32_t)pg.compressed_page_size, have = rd.size - body;
        if (have < sz) { code = P4S_E_PAGE_EOF; aux0 = sz; aux1 = have; break; }  // column_chunk_reader.cc:41-46
        if (n >= cd.max_pages) { code = E_OUT_CAPACITY; aux0 = cd.max_pages; break; }
        // the level-section prefixes of a v1 page that is stored as it is
        int32_t rep_prefix = -1, def_prefix = -1;
        if (pg.type == 0 && (pg.flags & P4S_PAGE_HAS_DATA_HEADER) && (cd.flags & P4S_CHUNK_UNCOMPRESSED)) {
            uint64_t p = body;
            bool ok = true;
            if (cd.flags & P4S_CHUNK_HAS_REP) {
                if (pg.rep_enc == 3) {                                            // RLE: encoding.cc:35-50
                    rep_prefix = hdr_prefix_at(rd, p, body + sz);
                    if (rep_prefix >= 0 && (uint64_t)rep_prefix + 4 <= sz) p += 4 + (uint64_t)rep_prefix; else ok = false;
                } else {
                    ok = false;                                                   // BIT_PACKED: the length depends on num_values and the level width
                }
            }
            if (ok && (cd.flags & P4S_CHUNK_HAS_DEF) && pg.def_enc == 3) def_prefix = hdr_prefix_at(rd, p, body + sz);
            rd.err = 0;
        }
        if (lane == 0) {
            p4s_page_info pi;
            pi.header_offset = at; pi.body_offset = body;
            pi.type = pg.type; pi.uncompressed_page_size = pg.uncompressed_page_size; pi.compressed_page_size = pg.compressed_page_size;
            pi.num_values = pg.num_values; pi.encoding = pg.encoding;
            pi.definition_level_encoding = pg.def_enc; pi.repetition_level_encoding = pg.rep_enc;
            pi.definition_levels_byte_length = pg.def_len; pi.repetition_levels_byte_length = pg.rep_len;
            pi.num_nulls = pg.num_nulls; pi.num_rows = pg.num_rows; pi.flags = pg.flags;
            pi.v1_rep_prefix = rep_prefix; pi.v1_def_prefix = def_prefix;
            pi.reserved[0] = 0; pi.reserved[1] = 0;
            P4S_ST(&out[n], pi, 82);
        }
        ++n;
        rd.seek(body + sz);
    }
    if (lane == 0) write_status(&status[c], code, n, aux0, aux1, code ? at : rd.pos, 0);
}

}  // namespace p4s
