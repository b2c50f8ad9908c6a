// Thin C driver around the reference's OWN header-only bit-level layer
// (include/parquet4seastar/rle_encoding.hh, bit_stream_utils.hh, bpacking.hh),
// which is std-only and compiles untouched from /root/reference.  Built only
// where /root/reference exists, into oracle/_ref/libp4s_ref_l0.so (see Makefile).
// TEST INFRASTRUCTURE ONLY: used to pin the CPU restatement in p4s_oracle.c.
//
// The page decoders of src/encoding.cc are NOT built here: they need Seastar
// and Thrift-generated headers that this image lacks.
#include <cassert>
#include <cstdint>
#include <cstring>
#include <parquet4seastar/rle_encoding.hh>

using parquet4seastar::RleDecoder;
using parquet4seastar::RleEncoder;
using parquet4seastar::BitUtil::BitReader;

namespace {

template <typename T>
void rle_batches(const uint8_t* buf, int len, int w, T* out, const int64_t* bs, int nb, int64_t* counts) {
    RleDecoder d(buf, len, w);
    for (int b = 0; b < nb; ++b) {
        counts[b] = d.GetBatch(out, static_cast<int>(bs[b]));
        out += counts[b];
    }
}

template <typename T>
int64_t bitpacked(const uint8_t* buf, int len, int w, T* out, int n) {
    BitReader r(buf, len);
    return r.GetBatch(w, out, n);
}

}  // namespace

extern "C" {

void ref_rle_decode_batches(const uint8_t* buf, int64_t len, int w, void* out, int elem_size,
                            const int64_t* bs, int nb, int64_t* counts) {
    switch (elem_size) {
    case 1: rle_batches(buf, (int)len, w, static_cast<uint8_t*>(out), bs, nb, counts); break;
    case 2: rle_batches(buf, (int)len, w, static_cast<int16_t*>(out), bs, nb, counts); break;
    case 4: rle_batches(buf, (int)len, w, static_cast<int32_t*>(out), bs, nb, counts); break;
    default: rle_batches(buf, (int)len, w, static_cast<uint64_t*>(out), bs, nb, counts); break;
    }
}

int64_t ref_bitpacked_decode(const uint8_t* buf, int64_t len, int w, void* out, int elem_size, int64_t n) {
    switch (elem_size) {
    case 1: return bitpacked(buf, (int)len, w, static_cast<uint8_t*>(out), (int)n);
    case 2: return bitpacked(buf, (int)len, w, static_cast<int16_t*>(out), (int)n);
    case 4: return bitpacked(buf, (int)len, w, static_cast<int32_t*>(out), (int)n);
    default: return bitpacked(buf, (int)len, w, static_cast<uint64_t*>(out), (int)n);
    }
}

// Scripted BitReader session.  ops[i] = {opcode, arg}:
//  0 GetValue<uint64_t>(arg bits)   1 GetAligned<uint64_t>(arg bytes)
//  2 GetVlqInt(u32)  3 GetVlqInt(u64)  4 GetZigZagVlqInt(i32)  5 GetZigZagVlqInt(i64)
//  6 bytes_left()
// results[i] = {ok, value}.
void ref_br_script(const uint8_t* buf, int64_t len, const int32_t* ops, int nops, int64_t* results) {
    BitReader r(buf, (int)len);
    for (int i = 0; i < nops; ++i) {
        int op = ops[2 * i], arg = ops[2 * i + 1];
        int64_t ok = 0, val = 0;
        switch (op) {
        case 0: { uint64_t v = 0; ok = r.GetValue(arg, &v); val = (int64_t)v; break; }
        case 1: { uint64_t v = 0; ok = r.GetAligned<uint64_t>(arg, &v); val = (int64_t)v; break; }
        case 2: { uint32_t v = 0; ok = r.GetVlqInt(&v); val = v; break; }
        case 3: { uint64_t v = 0; ok = r.GetVlqInt(&v); val = (int64_t)v; break; }
        case 4: { int32_t v = 0; ok = r.GetZigZagVlqInt(&v); val = v; break; }
        case 5: { int64_t v = 0; ok = r.GetZigZagVlqInt(&v); val = v; break; }
        default: ok = 1; val = r.bytes_left(); break;
        }
        results[2 * i] = ok;
        results[2 * i + 1] = ok ? val : 0;
    }
}

// Encode with the reference's RleEncoder (input generator for differential tests).
// Returns bytes written or -1 if the buffer was too small.
int64_t ref_rle_encode(const uint64_t* values, int64_t n, int w, uint8_t* out, int64_t cap) {
    if (cap < RleEncoder::MinBufferSize(w)) return -1;
    RleEncoder e(out, (int)cap, w);
    for (int64_t i = 0; i < n; ++i) {
        if (!e.Put(values[i])) return -1;
    }
    return e.Flush();
}

}  // extern "C"
