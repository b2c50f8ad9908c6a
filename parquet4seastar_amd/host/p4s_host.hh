// Host-side C++ mirror of the reference's page-decoding interface, backed by the gfx950
// kernels through the C ABI (include/p4s_gpu.h).  Class names, method signatures, argument
// meaning and exception texts follow the reference so that code written against
//   include/parquet4seastar/encoding.hh:42-82      (level_decoder)
//   include/parquet4seastar/encoding.hh:135-175    (decoder<T>, value_decoder<T>)
//   include/parquet4seastar/column_chunk_reader.hh (page_reader, column_chunk_reader<T>)
// ports by changing the namespace.  Differences, all forced by the absence of Seastar:
//   * read_batch returns size_t synchronously instead of seastar::future<size_t>
//     (adapter: seastar::make_ready_future<size_t>(r.read_batch(...)) or seastar::async);
//   * page_reader reads an in-memory column chunk instead of a seastar::input_stream;
//   * BYTE_ARRAY values are `bytes` (owning) instead of seastar::temporary_buffer<uint8_t>;
//   * only CompressionCodec::UNCOMPRESSED is accepted (decompression wraps external
//     libraries in the reference and is outside this path).
// There is no CPU decode path in this file: every value comes out of a kernel.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <exception>
#include <memory>
#include <optional>
#include <sstream>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/p4s_gpu.h"

namespace parquet4seastar_amd {

using byte = uint8_t;
using bytes = std::basic_string<byte>;
using bytes_view = std::basic_string_view<byte>;

// ref: include/parquet4seastar/exception.hh:42-56
class parquet_exception : public std::exception {
    std::string _msg;
public:
    static parquet_exception corrupted_file(const std::string& msg) {
        return parquet_exception("Invalid or corrupted parquet file: " + msg);
    }
    explicit parquet_exception(std::string msg) : _msg(std::move(msg)) {}
    const char* what() const noexcept override { return _msg.c_str(); }
};

template <typename... A>
inline std::string strcat_all(const A&... a) {
    std::ostringstream os;
    (os << ... << a);
    return os.str();
}

namespace format {
// values: include/parquet4seastar/parquet_types.h:24-35,90-102,110-120,128-135 of the reference
struct Type { enum type { BOOLEAN = 0, INT32 = 1, INT64 = 2, INT96 = 3, FLOAT = 4, DOUBLE = 5, BYTE_ARRAY = 6, FIXED_LEN_BYTE_ARRAY = 7 }; };
struct Encoding { enum type { PLAIN = 0, PLAIN_DICTIONARY = 2, RLE = 3, BIT_PACKED = 4, DELTA_BINARY_PACKED = 5,
                              DELTA_LENGTH_BYTE_ARRAY = 6, DELTA_BYTE_ARRAY = 7, RLE_DICTIONARY = 8, BYTE_STREAM_SPLIT = 9 }; };
struct CompressionCodec { enum type { UNCOMPRESSED = 0, SNAPPY = 1, GZIP = 2, LZO = 3, BROTLI = 4, LZ4 = 5, ZSTD = 6 }; };
struct PageType { enum type { DATA_PAGE = 0, INDEX_PAGE = 1, DICTIONARY_PAGE = 2, DATA_PAGE_V2 = 3 }; };

inline std::ostream& operator<<(std::ostream& os, Encoding::type e) {
    static const char* const names[] = {"PLAIN", nullptr, "PLAIN_DICTIONARY", "RLE", "BIT_PACKED", "DELTA_BINARY_PACKED",
                                        "DELTA_LENGTH_BYTE_ARRAY", "DELTA_BYTE_ARRAY", "RLE_DICTIONARY", "BYTE_STREAM_SPLIT"};
    int v = static_cast<int>(e);
    if (v >= 0 && v < 10 && names[v]) return os << names[v];
    return os << v;
}

struct DataPageHeader {
    int32_t num_values = 0;
    Encoding::type encoding = Encoding::PLAIN;
    Encoding::type definition_level_encoding = Encoding::RLE;
    Encoding::type repetition_level_encoding = Encoding::RLE;
};
struct DictionaryPageHeader {
    int32_t num_values = 0;
    Encoding::type encoding = Encoding::PLAIN;
    bool is_sorted = false;
};
struct DataPageHeaderV2 {
    int32_t num_values = 0, num_nulls = 0, num_rows = 0;
    Encoding::type encoding = Encoding::PLAIN;
    int32_t definition_levels_byte_length = 0, repetition_levels_byte_length = 0;
    bool is_compressed = true;
    bool isset_is_compressed = false;
};
struct PageHeader {
    PageType::type type = PageType::DATA_PAGE;
    int32_t uncompressed_page_size = 0, compressed_page_size = 0, crc = 0;
    DataPageHeader data_page_header;
    DictionaryPageHeader dictionary_page_header;
    DataPageHeaderV2 data_page_header_v2;
    struct { bool data_page_header = false, dictionary_page_header = false, data_page_header_v2 = false; } isset;
};
}  // namespace format

// ---------------------------------------------------------------------------
// Thrift compact protocol, just enough for PageHeader (doc/parquet/parquet.thrift:507-659 of the
// reference).  Unknown fields (Statistics, crc, index page headers) are skipped by wire type.
// ---------------------------------------------------------------------------
class thrift_compact_reader {
    const byte* _p;
    const byte* _end;
    const byte* _begin;
    [[noreturn]] void eof() const {
        throw parquet_exception(strcat_all("Could not deserialize thrift: unexpected end of stream at ", _end - _begin, "B"));
    }
    byte u8() { if (_p >= _end) eof(); return *_p++; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int sh = 0; sh < 70; sh += 7) {
            byte b = u8();
            v |= static_cast<uint64_t>(b & 0x7F) << sh;
            if (!(b & 0x80)) return v;
        }
        throw parquet_exception("Could not deserialize thrift: variable-length int over 10 bytes");
    }
    int64_t zigzag() { uint64_t u = varint(); return static_cast<int64_t>((u >> 1) ^ (~(u & 1) + 1)); }
    void skip_bytes(uint64_t n) { if (static_cast<uint64_t>(_end - _p) < n) eof(); _p += n; }
    void skip(int type, int depth = 0) {
        if (depth > 32) throw parquet_exception("Could not deserialize thrift: nesting too deep");
        switch (type) {
        case 1: case 2: break;
        case 3: u8(); break;
        case 4: case 5: case 6: varint(); break;
        case 7: skip_bytes(8); break;
        case 8: skip_bytes(varint()); break;
        case 9: case 10: {
            byte h = u8();
            uint64_t n = h >> 4;
            if (n == 15) n = varint();
            for (uint64_t i = 0; i < n; ++i) skip_elem(h & 0x0F, depth + 1);
            break;
        }
        case 11: {
            uint64_t n = varint();
            if (n) { byte kv = u8(); for (uint64_t i = 0; i < n; ++i) { skip_elem(kv >> 4, depth + 1); skip_elem(kv & 0x0F, depth + 1); } }
            break;
        }
        case 12: skip_struct(depth + 1); break;
        default: throw parquet_exception(strcat_all("Could not deserialize thrift: invalid wire type ", type));
        }
    }
    void skip_elem(int type, int depth) { if (type == 1 || type == 2) u8(); else skip(type, depth); }
    void skip_struct(int depth) {
        int16_t last = 0;
        for (;;) { int id, type; if (!field(last, id, type)) return; skip(type, depth); }
    }
    // next field header of the current struct; false at STOP
    bool field(int16_t& last, int& id, int& type) {
        byte h = u8();
        if (h == 0) return false;
        type = h & 0x0F;
        int delta = h >> 4;
        id = delta ? last + delta : static_cast<int>(zigzag());
        last = static_cast<int16_t>(id);
        return true;
    }
    int32_t i32(int type) {
        if (type != 5 && type != 4 && type != 6) throw parquet_exception("Could not deserialize thrift: expected an integer field");
        return static_cast<int32_t>(zigzag());
    }
public:
    thrift_compact_reader(const byte* p, size_t n) : _p(p), _end(p + n), _begin(p) {}
    size_t consumed() const { return static_cast<size_t>(_p - _begin); }

    void read(format::DataPageHeader& h) {
        int16_t last = 0; int id, type;
        while (field(last, id, type)) {
            switch (id) {
            case 1: h.num_values = i32(type); break;
            case 2: h.encoding = static_cast<format::Encoding::type>(i32(type)); break;
            case 3: h.definition_level_encoding = static_cast<format::Encoding::type>(i32(type)); break;
            case 4: h.repetition_level_encoding = static_cast<format::Encoding::type>(i32(type)); break;
            default: skip(type);
            }
        }
    }
    void read(format::DictionaryPageHeader& h) {
        int16_t last = 0; int id, type;
        while (field(last, id, type)) {
            switch (id) {
            case 1: h.num_values = i32(type); break;
            case 2: h.encoding = static_cast<format::Encoding::type>(i32(type)); break;
            case 3: h.is_sorted = (type == 1); break;
            default: skip(type);
            }
        }
    }
    void read(format::DataPageHeaderV2& h) {
        int16_t last = 0; int id, type;
        while (field(last, id, type)) {
            switch (id) {
            case 1: h.num_values = i32(type); break;
            case 2: h.num_nulls = i32(type); break;
            case 3: h.num_rows = i32(type); break;
            case 4: h.encoding = static_cast<format::Encoding::type>(i32(type)); break;
            case 5: h.definition_levels_byte_length = i32(type); break;
            case 6: h.repetition_levels_byte_length = i32(type); break;
            case 7: h.is_compressed = (type == 1); h.isset_is_compressed = true; break;
            default: skip(type);
            }
        }
    }
    void read(format::PageHeader& h) {
        int16_t last = 0; int id, type;
        while (field(last, id, type)) {
            switch (id) {
            case 1: h.type = static_cast<format::PageType::type>(i32(type)); break;
            case 2: h.uncompressed_page_size = i32(type); break;
            case 3: h.compressed_page_size = i32(type); break;
            case 4: h.crc = i32(type); break;
            case 5: if (type != 12) skip(type); else { read(h.data_page_header); h.isset.data_page_header = true; } break;
            case 7: if (type != 12) skip(type); else { read(h.dictionary_page_header); h.isset.dictionary_page_header = true; } break;
            case 8: if (type != 12) skip(type); else { read(h.data_page_header_v2); h.isset.data_page_header_v2 = true; } break;
            default: skip(type);
            }
        }
    }
};

// Serializer used by tests and tools that assemble column chunks (the reference writes page
// headers with Thrift's TCompactProtocol, src/thrift_serdes.hh:96-113).
class thrift_compact_writer {
    bytes _out;
    void varint(uint64_t v) { while (v >= 0x80) { _out.push_back(static_cast<byte>(v | 0x80)); v >>= 7; } _out.push_back(static_cast<byte>(v)); }
    void zz(int64_t v) { varint((static_cast<uint64_t>(v) << 1) ^ static_cast<uint64_t>(v >> 63)); }
    void fld(int16_t& last, int id, int type) {
        int d = id - last;
        if (d > 0 && d <= 15) _out.push_back(static_cast<byte>((d << 4) | type));
        else { _out.push_back(static_cast<byte>(type)); zz(id); }
        last = static_cast<int16_t>(id);
    }
    void i32f(int16_t& last, int id, int32_t v) { fld(last, id, 5); zz(v); }
public:
    const bytes& serialize(const format::PageHeader& h) {
        _out.clear();
        int16_t last = 0;
        i32f(last, 1, h.type);
        i32f(last, 2, h.uncompressed_page_size);
        i32f(last, 3, h.compressed_page_size);
        if (h.isset.data_page_header) {
            fld(last, 5, 12);
            int16_t l = 0;
            i32f(l, 1, h.data_page_header.num_values);
            i32f(l, 2, h.data_page_header.encoding);
            i32f(l, 3, h.data_page_header.definition_level_encoding);
            i32f(l, 4, h.data_page_header.repetition_level_encoding);
            _out.push_back(0);
        }
        if (h.isset.dictionary_page_header) {
            fld(last, 7, 12);
            int16_t l = 0;
            i32f(l, 1, h.dictionary_page_header.num_values);
            i32f(l, 2, h.dictionary_page_header.encoding);
            _out.push_back(0);
        }
        if (h.isset.data_page_header_v2) {
            fld(last, 8, 12);
            int16_t l = 0;
            const auto& v = h.data_page_header_v2;
            i32f(l, 1, v.num_values); i32f(l, 2, v.num_nulls); i32f(l, 3, v.num_rows); i32f(l, 4, v.encoding);
            i32f(l, 5, v.definition_levels_byte_length); i32f(l, 6, v.repetition_levels_byte_length);
            if (v.isset_is_compressed) fld(l, 7, v.is_compressed ? 1 : 2);
            _out.push_back(0);
        }
        _out.push_back(0);
        return _out;
    }
};

// ---------------------------------------------------------------------------
// Device plumbing: one context per thread, grow-only staging buffers.
// ---------------------------------------------------------------------------
class device {
    p4s_ctx* _ctx = nullptr;
    struct slot { void* p = nullptr; uint64_t cap = 0; };
    slot _in, _out, _aux, _desc, _status, _scratch, _dict;
    void check(int rc, const char* what) {
        if (rc != P4S_OK) throw std::runtime_error(strcat_all(what, " failed (", rc, "): ", p4s_last_error(_ctx)));
    }
    void* ensure(slot& s, uint64_t n) {
        if (n > s.cap) {
            if (s.p) check(p4s_dev_free(_ctx, s.p), "p4s_dev_free");
            s.p = nullptr; s.cap = 0;
            uint64_t want = std::max<uint64_t>(n + n / 2, 4096);
            check(p4s_dev_alloc(_ctx, want, &s.p), "p4s_dev_alloc");
            s.cap = want;
        }
        return s.p;
    }
public:
    explicit device(int index) {
        int rc = p4s_ctx_create(index, &_ctx);
        if (rc != P4S_OK) throw std::runtime_error(strcat_all("p4s_ctx_create(", index, ") failed (", rc, "): no usable GPU"));
    }
    ~device() {
        for (slot* s : {&_in, &_out, &_aux, &_desc, &_status, &_scratch, &_dict}) if (s->p) p4s_dev_free(_ctx, s->p);
        p4s_ctx_destroy(_ctx);
    }
    device(const device&) = delete;
    device& operator=(const device&) = delete;
    static device& current() {
        static thread_local std::unique_ptr<device> d;
        if (!d) {
            const char* e = std::getenv("P4S_DEVICE");
            d = std::make_unique<device>(e ? std::atoi(e) : 0);
        }
        return *d;
    }
    p4s_ctx* ctx() { return _ctx; }

    // What one page produced.
    struct result {
        p4s_page_status st{};
        [[noreturn]] void raise() const {
            char buf[512];
            p4s_format_error(&st, buf, sizeof buf);
            throw parquet_exception(std::string(buf));
        }
    };

    enum class op { plain_fixed, plain_bool, bss, delta, rle, levels, dict, dlba, dba, plain_ba, flba };

    // Fixed-width output: decodes one page, copies min(values_decoded, cap) elements to `out`.
    result run_fixed(op o, bytes_view page, uint32_t n, uint32_t p0, uint32_t p1, uint32_t flags, int elem,
                     void* out, uint64_t cap, const void* h_dict = nullptr, uint64_t dict_size = 0) {
        uint8_t* d_in = static_cast<uint8_t*>(ensure(_in, page.size() + 16));
        void* d_out = ensure(_out, std::max<uint64_t>(cap, 1) * elem);
        auto* d_desc = static_cast<p4s_page_desc*>(ensure(_desc, sizeof(p4s_page_desc)));
        auto* d_st = static_cast<p4s_page_status*>(ensure(_status, sizeof(p4s_page_status)));
        p4s_page_desc pd{};
        pd.in_offset = 0; pd.in_size = page.size(); pd.out_offset = 0; pd.aux_offset = 0;
        pd.num_values = static_cast<uint32_t>(std::min<uint64_t>(n, cap));
        pd.param0 = p0; pd.param1 = p1; pd.flags = flags;
        check(p4s_copy_h2d(_ctx, d_in, page.data(), page.size()), "h2d(page)");
        check(p4s_copy_h2d(_ctx, d_desc, &pd, sizeof pd), "h2d(desc)");
        const uint64_t inb = page.size();
        int rc = P4S_OK;
        switch (o) {
        case op::plain_fixed: rc = p4s_decode_plain_fixed(_ctx, d_in, inb, d_desc, 1, d_out, d_st, elem); break;
        case op::plain_bool: rc = p4s_decode_plain_boolean(_ctx, d_in, inb, d_desc, 1, static_cast<uint8_t*>(d_out), d_st); break;
        case op::bss: rc = p4s_decode_byte_stream_split(_ctx, d_in, inb, d_desc, 1, d_out, d_st, elem); break;
        case op::delta: rc = p4s_decode_delta_binary_packed(_ctx, d_in, inb, d_desc, 1, d_out, d_st, elem); break;
        case op::rle: rc = p4s_decode_rle(_ctx, d_in, inb, d_desc, 1, d_out, d_st, elem); break;
        case op::levels: rc = p4s_decode_levels(_ctx, d_in, inb, d_desc, 1, d_out, d_st, elem); break;
        case op::dict: {
            void* d_dict = ensure(_dict, (dict_size + 1) * elem);
            check(p4s_copy_h2d(_ctx, d_dict, h_dict, (dict_size + 1) * elem), "h2d(dict)");
            rc = p4s_decode_dictionary(_ctx, d_in, inb, d_desc, 1, d_dict, dict_size, elem, d_out, d_st);
            break;
        }
        default: throw std::logic_error("run_fixed: not a fixed-width decoder");
        }
        check(rc, "decode");
        result r;
        check(p4s_copy_d2h(_ctx, &r.st, d_st, sizeof r.st), "d2h(status)");
        check(p4s_ctx_sync(_ctx), "sync");
        if (r.st.code == 0 && r.st.values_decoded) {
            check(p4s_copy_d2h(_ctx, out, d_out, std::min<uint64_t>(r.st.values_decoded, cap) * elem), "d2h(values)");
            check(p4s_ctx_sync(_ctx), "sync");
        }
        return r;
    }

    // (offsets, bytes) output.  DELTA_BYTE_ARRAY runs the size pass first.
    result run_strings(op o, bytes_view page, uint32_t n, uint32_t flba_len, std::vector<int64_t>& offsets, std::vector<byte>& data) {
        uint8_t* d_in = static_cast<uint8_t*>(ensure(_in, page.size() + 16));
        auto* d_desc = static_cast<p4s_page_desc*>(ensure(_desc, sizeof(p4s_page_desc)));
        auto* d_st = static_cast<p4s_page_status*>(ensure(_status, sizeof(p4s_page_status)));
        const uint64_t inb = page.size();
        // a page cannot hold more strings than bytes (+1 for the all-empty corner cases of w=0 lengths)
        uint64_t max_vals = std::min<uint64_t>(n, o == op::flba ? n : std::max<uint64_t>(inb * 8 + 1, 1));
        if (o == op::dlba || o == op::dba) max_vals = std::min<uint64_t>(n, delta_total(page));
        auto* d_off = static_cast<int64_t*>(ensure(_out, (max_vals + 2) * 8));
        p4s_page_desc pd{};
        pd.in_size = inb; pd.num_values = static_cast<uint32_t>(max_vals);
        check(p4s_copy_h2d(_ctx, d_in, page.data(), page.size()), "h2d(page)");
        result r;
        uint64_t cap = inb;
        if (o == op::flba) { pd.param0 = flba_len; cap = inb; }
        if (o == op::dba) {
            void* d_scr = ensure(_scratch, (max_vals + 2) * 8);
            check(p4s_copy_h2d(_ctx, d_desc, &pd, sizeof pd), "h2d(desc)");
            check(p4s_delta_byte_array_sizes(_ctx, d_in, inb, d_desc, 1, d_scr, (max_vals + 2) * 8, d_st), "dba sizes");
            check(p4s_copy_d2h(_ctx, &r.st, d_st, sizeof r.st), "d2h(status)");
            check(p4s_ctx_sync(_ctx), "sync");
            if (r.st.code != 0) return r;
            cap = r.st.out_bytes;
        }
        if (o != op::flba) pd.param0 = static_cast<uint32_t>(std::min<uint64_t>(cap, 0xFFFFFFFFull));
        uint8_t* d_bytes = static_cast<uint8_t*>(ensure(_aux, cap + 16));
        check(p4s_copy_h2d(_ctx, d_desc, &pd, sizeof pd), "h2d(desc)");
        int rc = P4S_OK;
        switch (o) {
        case op::dlba: rc = p4s_decode_delta_length_byte_array(_ctx, d_in, inb, d_desc, 1, d_off, d_bytes, cap, d_st); break;
        case op::plain_ba: rc = p4s_decode_plain_byte_array(_ctx, d_in, inb, d_desc, 1, d_off, d_bytes, cap, d_st); break;
        case op::flba: rc = p4s_decode_plain_fixed_len_byte_array(_ctx, d_in, inb, d_desc, 1, d_bytes, cap, d_st); break;
        case op::dba: rc = p4s_decode_delta_byte_array(_ctx, d_in, inb, d_desc, 1, d_off, d_bytes, cap, _scratch.p, (max_vals + 2) * 8, d_st); break;
        default: throw std::logic_error("run_strings: not a string decoder");
        }
        check(rc, "decode");
        check(p4s_copy_d2h(_ctx, &r.st, d_st, sizeof r.st), "d2h(status)");
        check(p4s_ctx_sync(_ctx), "sync");
        if (r.st.code != 0) return r;
        const uint64_t nv = r.st.values_decoded, nb = r.st.out_bytes;
        offsets.assign(nv + 1, 0);
        data.resize(nb);
        if (o == op::flba) {
            for (uint64_t i = 0; i <= nv; ++i) offsets[i] = static_cast<int64_t>(i * flba_len);
        } else {
            check(p4s_copy_d2h(_ctx, offsets.data(), d_off, (nv + 1) * 8), "d2h(offsets)");
        }
        if (nb) check(p4s_copy_d2h(_ctx, data.data(), d_bytes, nb), "d2h(bytes)");
        check(p4s_ctx_sync(_ctx), "sync");
        return r;
    }

    // value count announced by a DELTA_BINARY_PACKED header (third ULEB128), 2^32-1 when unreadable
    static uint64_t delta_total(bytes_view page) {
        size_t pos = 0;
        uint64_t v = 0;
        for (int f = 0; f < 3; ++f) {
            v = 0;
            int sh = 0;
            for (;;) {
                if (pos >= page.size() || sh > 63) return 0xFFFFFFFFull;
                byte b = page[pos++];
                v |= static_cast<uint64_t>(b & 0x7F) << sh;
                sh += 7;
                if (!(b & 0x80)) break;
            }
        }
        return v;
    }
};

// ref: include/parquet4seastar/encoding.hh:35-37
constexpr inline uint32_t bit_width(uint64_t max_n) {
    uint32_t w = 0;
    while (max_n) { ++w; max_n >>= 1; }
    return w;
}

// ---------------------------------------------------------------------------
// level_decoder (ref: include/parquet4seastar/encoding.hh:42-82, src/encoding.cc:26-79).
// reset_* validates the framing on the host (the caller needs the consumed size at once) and
// keeps a copy of the section; the levels are decoded on the GPU, all num_values of them, at
// the first read_batch.
// ---------------------------------------------------------------------------
class level_decoder {
    uint32_t _max_level;
    uint32_t _bit_width;
    uint32_t _num_values = 0;
    uint32_t _values_read = 0;
    uint32_t _flags = P4S_LEVELS_V2;
    bytes _section;
    bool _decoded = false;
    std::vector<int32_t> _levels;
    uint64_t _hits = 0;
    void decode() {
        _levels.assign(std::max<uint32_t>(_num_values, 1), 0);
        auto r = device::current().run_fixed(device::op::levels, _section, _num_values, _max_level, _num_values, _flags, 4,
                                             _levels.data(), _num_values);
        if (r.st.code != 0) r.raise();
        _levels.resize(r.st.values_decoded);
        _hits = r.st.aux0;
        _decoded = true;
    }
public:
    explicit level_decoder(uint32_t max_level) : _max_level(max_level), _bit_width(bit_width(max_level)) {}

    size_t reset_v1(bytes_view buffer, format::Encoding::type encoding, uint32_t num_values) {
        _num_values = num_values;
        _values_read = 0;
        _decoded = false;
        if (_bit_width == 0) { _section.clear(); _flags = P4S_LEVELS_V1_RLE; return 0; }
        if (encoding == format::Encoding::RLE) {
            if (buffer.size() < 4)
                throw parquet_exception::corrupted_file(strcat_all("End of page while reading levels (needed ", 4, "B, got ", buffer.size(), "B)"));
            int32_t len;
            std::memcpy(&len, buffer.data(), 4);
            if (len < 0) throw parquet_exception::corrupted_file(strcat_all("Negative RLE levels length (", len, ")"));
            if (static_cast<size_t>(len) > buffer.size())
                throw parquet_exception::corrupted_file(strcat_all("End of page while reading levels (needed ", len, "B, got ", buffer.size(), "B)"));
            size_t used = std::min<size_t>(4 + static_cast<size_t>(len), buffer.size());   // the reference may run 4 bytes past the page here
            _section.assign(buffer.data(), used);
            _flags = P4S_LEVELS_V1_RLE;
            return used;
        } else if (encoding == format::Encoding::BIT_PACKED) {
            uint64_t byte_len = (static_cast<uint64_t>(num_values) * _bit_width + 7) >> 3;
            if (byte_len > 0x7FFFFFFFull)
                throw parquet_exception::corrupted_file(strcat_all("BIT_PACKED length exceeds int (", byte_len, "B)"));
            if (byte_len > buffer.size())
                throw parquet_exception::corrupted_file(strcat_all("End of page while reading levels (needed ", byte_len, "B, got ", buffer.size(), "B)"));
            _section.assign(buffer.data(), byte_len);
            _flags = P4S_LEVELS_V1_BIT_PACKED;
            return byte_len;
        }
        throw parquet_exception(strcat_all("Unknown level encoding (", encoding, ")"));
    }

    void reset_v2(bytes_view encoded_levels, uint32_t num_values) {
        _num_values = num_values;
        _values_read = 0;
        _decoded = false;
        if (encoded_levels.size() > 0x7FFFFFFFull)
            throw parquet_exception::corrupted_file(strcat_all("Levels length exceeds int (", encoded_levels.size(), "B)"));
        _section.assign(encoded_levels.data(), encoded_levels.size());
        _flags = P4S_LEVELS_V2;
    }

    template <typename T>
    uint32_t read_batch(uint32_t n, T out[]) {
        n = std::min(n, _num_values - _values_read);
        if (_bit_width == 0) {
            std::fill(out, out + n, 0);
            _values_read += n;
            return n;
        }
        if (!_decoded) decode();
        const uint32_t avail = static_cast<uint32_t>(_levels.size()) > _values_read ? static_cast<uint32_t>(_levels.size()) - _values_read : 0;
        n = std::min(n, avail);
        for (uint32_t i = 0; i < n; ++i) out[i] = static_cast<T>(_levels[_values_read + i]);
        _values_read += n;
        return n;
    }
};

// ---------------------------------------------------------------------------
// value decoders (ref: include/parquet4seastar/encoding.hh:84-175, src/encoding.cc:81-616)
// ---------------------------------------------------------------------------
template <format::Type::type T> struct value_decoder_traits;
template <> struct value_decoder_traits<format::Type::INT32> { using output_type = int32_t; };
template <> struct value_decoder_traits<format::Type::INT64> { using output_type = int64_t; };
template <> struct value_decoder_traits<format::Type::INT96> { using output_type = std::array<int32_t, 3>; };
template <> struct value_decoder_traits<format::Type::FLOAT> { using output_type = float; };
template <> struct value_decoder_traits<format::Type::DOUBLE> { using output_type = double; };
template <> struct value_decoder_traits<format::Type::BOOLEAN> { using output_type = uint8_t; };
template <> struct value_decoder_traits<format::Type::BYTE_ARRAY> { using output_type = bytes; };
template <> struct value_decoder_traits<format::Type::FIXED_LEN_BYTE_ARRAY> { using output_type = bytes; };

template <format::Type::type ParquetType>
class decoder {
public:
    using output_type = typename value_decoder_traits<ParquetType>::output_type;
    virtual void reset_dict(output_type*, size_t) {}
    virtual void reset(bytes_view buf) = 0;
    virtual size_t read_batch(size_t n, output_type out[]) = 0;
    virtual ~decoder() = default;
};

// One decoder class serves every encoding: it keeps the page, and decodes on the GPU as many
// values as the calls so far have asked for (one kernel pass for the usual whole-page call;
// a later, larger request re-runs the page with a larger n: valid streams are batch-invariant).
template <format::Type::type ParquetType>
class gpu_page_decoder final : public decoder<ParquetType> {
public:
    using output_type = typename decoder<ParquetType>::output_type;
    static constexpr bool is_string = ParquetType == format::Type::BYTE_ARRAY || ParquetType == format::Type::FIXED_LEN_BYTE_ARRAY;
private:
    format::Encoding::type _encoding;
    uint32_t _type_length;
    output_type* _dict;
    size_t _dict_size;
    bytes _page;
    size_t _served = 0;
    size_t _asked = 0;          // n of the last GPU pass
    bool _exhausted = false;    // the last pass returned fewer values than asked
    std::vector<output_type> _values;

    static constexpr int elem() {
        if constexpr (is_string) return 4; else return static_cast<int>(sizeof(output_type));
    }
    bool dict_encoded() const { return _encoding == format::Encoding::RLE_DICTIONARY || _encoding == format::Encoding::PLAIN_DICTIONARY; }

    // upper bound of what the page can hold, to size the output without trusting n
    uint64_t bound(uint64_t n) const {
        const uint64_t sz = _page.size();
        switch (_encoding) {
        case format::Encoding::PLAIN:
            if constexpr (ParquetType == format::Type::BOOLEAN) return std::min<uint64_t>(n, sz * 8);
            else if constexpr (is_string) return n;
            else return std::min<uint64_t>(n, sz / sizeof(output_type));
        case format::Encoding::BYTE_STREAM_SPLIT: return std::min<uint64_t>(n, sz / elem());
        case format::Encoding::DELTA_BINARY_PACKED: return std::min<uint64_t>(n, device::delta_total(_page));
        default: return std::min<uint64_t>(n, 1ull << 28);
        }
    }

    void run(uint64_t n) {
        device& dev = device::current();
        n = std::min<uint64_t>(n, 0xFFFFFFFFull);
        const uint64_t cap = bound(n);
        device::result r;
        if constexpr (is_string) {
            std::vector<int64_t> off;
            std::vector<byte> data;
            if (dict_encoded()) {
                // indices through the identity dictionary {0..dict_size}, strings gathered here
                std::vector<uint32_t> ident(_dict_size + 1), idx(std::max<uint64_t>(cap, 1));
                for (size_t i = 0; i <= _dict_size; ++i) ident[i] = static_cast<uint32_t>(i);
                r = dev.run_fixed(device::op::dict, _page, static_cast<uint32_t>(n), 0, 0, 0, 4, idx.data(), cap, ident.data(), _dict_size);
                if (r.st.code != 0) r.raise();
                _values.clear();
                for (uint64_t i = 0; i < r.st.values_decoded; ++i)
                    _values.push_back(idx[i] < _dict_size ? _dict[idx[i]] : output_type());
            } else {
                device::op o = _encoding == format::Encoding::PLAIN
                                   ? (ParquetType == format::Type::BYTE_ARRAY ? device::op::plain_ba : device::op::flba)
                                   : (_encoding == format::Encoding::DELTA_LENGTH_BYTE_ARRAY ? device::op::dlba : device::op::dba);
                r = dev.run_strings(o, _page, static_cast<uint32_t>(n), _type_length, off, data);
                if (r.st.code != 0) r.raise();
                _values.clear();
                for (uint64_t i = 0; i < r.st.values_decoded; ++i)
                    _values.emplace_back(data.data() + off[i], static_cast<size_t>(off[i + 1] - off[i]));
            }
        } else {
            _values.assign(std::max<uint64_t>(cap, 1), output_type());
            device::op o;
            uint32_t p0 = 0;
            switch (_encoding) {
            case format::Encoding::PLAIN: o = ParquetType == format::Type::BOOLEAN ? device::op::plain_bool : device::op::plain_fixed; break;
            case format::Encoding::RLE: o = device::op::rle; p0 = 1; break;
            case format::Encoding::DELTA_BINARY_PACKED: o = device::op::delta; break;
            case format::Encoding::BYTE_STREAM_SPLIT: o = device::op::bss; break;
            default: o = device::op::dict; break;
            }
            if (o == device::op::dict) {
                std::vector<output_type> padded(_dict_size + 1);
                std::copy(_dict, _dict + _dict_size, padded.begin());
                r = dev.run_fixed(o, _page, static_cast<uint32_t>(n), 0, 0, 0, elem(), _values.data(), cap, padded.data(), _dict_size);
            } else {
                r = dev.run_fixed(o, _page, static_cast<uint32_t>(n), p0, 0, 0, elem(), _values.data(), cap);
            }
            if (r.st.code != 0) r.raise();
            _values.resize(std::min<uint64_t>(r.st.values_decoded, cap));
        }
        _asked = n;
        _exhausted = r.st.values_decoded < n;
    }
public:
    gpu_page_decoder(format::Encoding::type encoding, uint32_t type_length, output_type* dict, size_t dict_size)
        : _encoding(encoding), _type_length(type_length), _dict(dict), _dict_size(dict_size) {}

    void reset(bytes_view data) override {
        if (_encoding == format::Encoding::BYTE_STREAM_SPLIT && data.size() % sizeof(output_type) != 0)
            throw parquet_exception("Data size in BYTE_STREAM_SPLIT is not divisible by size of data type");   // ref :386-389
        if (dict_encoded()) {
            if (data.size() == 0) throw parquet_exception::corrupted_file("Empty dictionary-encoded data page");
            if (data[0] > 32)                                                                                // ref :473-477
                throw parquet_exception::corrupted_file(strcat_all(
                    "Illegal dictionary index bit width (should be 0 <= bit width <= 32, got ", static_cast<int>(data[0]), ")"));
        }
        if (_encoding == format::Encoding::DELTA_BINARY_PACKED || _encoding == format::Encoding::DELTA_LENGTH_BYTE_ARRAY ||
            _encoding == format::Encoding::DELTA_BYTE_ARRAY) {
            // the reference parses the stream header (and, for the byte-array forms, every length) in reset();
            // run the page now so that those errors surface from reset() as well
            _page.assign(data.data(), data.size());
            _served = 0;
            run(0xFFFFFFFFull);
            return;
        }
        _page.assign(data.data(), data.size());
        _served = 0; _asked = 0; _exhausted = false;
        _values.clear();
    }

    size_t read_batch(size_t n, output_type out[]) override {
        if (_served + n > _values.size() && !_exhausted && _served + n > _asked)
            run(std::max<uint64_t>(_served + n, 2 * _asked));
        const size_t avail = _values.size() > _served ? _values.size() - _served : 0;
        n = std::min(n, avail);
        for (size_t i = 0; i < n; ++i) out[i] = _values[_served + i];
        _served += n;
        return n;
    }
};

// ref: include/parquet4seastar/encoding.hh:150-175, src/encoding.cc:519-591
template <format::Type::type ParquetType>
class value_decoder {
public:
    using output_type = typename value_decoder_traits<ParquetType>::output_type;
private:
    std::unique_ptr<decoder<ParquetType>> _decoder;
    std::optional<uint32_t> _type_length;
    bool _dict_set = false;
    output_type* _dict = nullptr;
    size_t _dict_size = 0;
public:
    value_decoder(std::optional<uint32_t> type_length) : _type_length(type_length) {
        if constexpr (ParquetType == format::Type::FIXED_LEN_BYTE_ARRAY) {
            if (!_type_length) throw parquet_exception::corrupted_file("type_length not set for FIXED_LEN_BYTE_ARRAY");
        }
    }
    void reset_dict(output_type* dictionary, size_t dictionary_size) {
        _dict = dictionary;
        _dict_size = dictionary_size;
        _dict_set = true;
    }
    void reset(bytes_view buf, format::Encoding::type encoding) {
        using T = format::Type;
        using E = format::Encoding;
        switch (encoding) {
        case E::PLAIN: break;
        case E::RLE_DICTIONARY:
        case E::PLAIN_DICTIONARY:
            if (!_dict_set) throw parquet_exception::corrupted_file("No dictionary page found before a dictionary-encoded page");
            if constexpr (ParquetType == T::BOOLEAN) throw parquet_exception("Dictionary-encoded BOOLEAN pages are not supported on the GPU path");
            break;
        case E::RLE:
            if constexpr (ParquetType != T::BOOLEAN) throw parquet_exception::corrupted_file("RLE encoding is valid only for BOOLEAN values");
            break;
        case E::DELTA_BINARY_PACKED:
            if constexpr (ParquetType != T::INT32 && ParquetType != T::INT64)
                throw parquet_exception::corrupted_file("DELTA_BINARY_PACKED is valid only for INT32 and INT64");
            break;
        case E::DELTA_LENGTH_BYTE_ARRAY:
            if constexpr (ParquetType != T::BYTE_ARRAY) throw parquet_exception::corrupted_file("DELTA_LENGTH_BYTE_ARRAY is valid only for BYTE_ARRAY");
            break;
        case E::DELTA_BYTE_ARRAY:
            if constexpr (ParquetType != T::BYTE_ARRAY) throw parquet_exception::corrupted_file("DELTA_BYTE_ARRAY is valid only for BYTE_ARRAY");
            break;
        case E::BYTE_STREAM_SPLIT:
            if constexpr (ParquetType != T::FLOAT && ParquetType != T::DOUBLE)
                throw parquet_exception::corrupted_file("BYTE_STREAM_SPLIT is valid only for FLOAT and DOUBLE");
            break;
        default:
            throw parquet_exception(strcat_all("Encoding ", encoding, " not implemented"));
        }
        _decoder = std::make_unique<gpu_page_decoder<ParquetType>>(encoding, _type_length.value_or(0), _dict, _dict_size);
        _decoder->reset(buf);
    }
    size_t read_batch(size_t n, output_type out[]) { return _decoder->read_batch(n, out); }
};

// ---------------------------------------------------------------------------
// page_reader / column_chunk_reader (ref: include/parquet4seastar/column_chunk_reader.hh:31-157,
// src/column_chunk_reader.cc:27-163)
// ---------------------------------------------------------------------------
struct page {
    const format::PageHeader* header;
    bytes_view contents;
};

class page_reader {
    bytes _chunk;
    size_t _pos = 0;
    std::unique_ptr<format::PageHeader> _latest_header = std::make_unique<format::PageHeader>();
public:
    explicit page_reader(bytes chunk) : _chunk(std::move(chunk)) {}
    explicit page_reader(bytes_view chunk) : _chunk(chunk) {}
    // View the next page. Returns an empty result on eof.
    std::optional<page> next_page() {
        *_latest_header = format::PageHeader{};
        if (_pos >= _chunk.size()) return std::nullopt;
        thrift_compact_reader rd(_chunk.data() + _pos, _chunk.size() - _pos);
        rd.read(*_latest_header);
        _pos += rd.consumed();
        if (_latest_header->compressed_page_size < 0)
            throw parquet_exception::corrupted_file(strcat_all("Negative compressed_page_size in header: ", _latest_header->compressed_page_size));
        const size_t sz = static_cast<uint32_t>(_latest_header->compressed_page_size);
        const size_t have = _chunk.size() - _pos;
        if (have < sz)
            throw parquet_exception::corrupted_file(strcat_all(
                "Unexpected end of column chunk while reading compressed page contents (expected ", sz, "B, got ", have, "B)"));
        bytes_view contents(_chunk.data() + _pos, sz);
        _pos += sz;
        return page{_latest_header.get(), contents};
    }
};

template <format::Type::type T>
class column_chunk_reader {
public:
    using output_type = typename value_decoder_traits<T>::output_type;
private:
    page_reader _source;
    level_decoder _rep_decoder;
    level_decoder _def_decoder;
    value_decoder<T> _val_decoder;
    std::optional<std::vector<output_type>> _dict;
    bool _initialized = false;
    bool _eof = false;
    int64_t _page_ordinal = -1;
    uint32_t _def_level;
    uint32_t _rep_level;
    std::optional<uint32_t> _type_length;

    void load_data_page(page p) {                                   // ref: column_chunk_reader.cc:52-78
        if (!p.header->isset.data_page_header) throw parquet_exception::corrupted_file("DataPageHeader not set for DATA_PAGE header");
        const format::DataPageHeader& header = p.header->data_page_header;
        if (header.num_values < 0) throw parquet_exception::corrupted_file(strcat_all("Negative num_values in header: ", header.num_values));
        if (p.header->uncompressed_page_size < 0)
            throw parquet_exception::corrupted_file(strcat_all("Negative uncompressed_page_size in header: ", p.header->uncompressed_page_size));
        bytes_view contents = p.contents;
        size_t n_read = _rep_decoder.reset_v1(contents, header.repetition_level_encoding, header.num_values);
        contents.remove_prefix(n_read);
        n_read = _def_decoder.reset_v1(contents, header.definition_level_encoding, header.num_values);
        contents.remove_prefix(n_read);
        _val_decoder.reset(contents, header.encoding);
    }
    void load_data_page_v2(page p) {                                // ref: column_chunk_reader.cc:80-111
        if (!p.header->isset.data_page_header_v2) throw parquet_exception::corrupted_file("DataPageHeaderV2 not set for DATA_PAGE_V2 header");
        const format::DataPageHeaderV2& header = p.header->data_page_header_v2;
        if (header.num_values < 0) throw parquet_exception::corrupted_file(strcat_all("Negative num_values in header: ", header.num_values));
        if (header.repetition_levels_byte_length < 0 || header.definition_levels_byte_length < 0)
            throw parquet_exception::corrupted_file("Negative levels byte length in header");
        if (p.header->uncompressed_page_size < 0)
            throw parquet_exception::corrupted_file(strcat_all("Negative uncompressed_page_size in header: ", p.header->uncompressed_page_size));
        bytes_view contents = p.contents;
        const size_t rl = static_cast<size_t>(header.repetition_levels_byte_length), dl = static_cast<size_t>(header.definition_levels_byte_length);
        if (rl + dl > contents.size()) throw parquet_exception::corrupted_file("Levels byte length exceeds the page");
        _rep_decoder.reset_v2(contents.substr(0, rl), header.num_values);
        contents.remove_prefix(rl);
        _def_decoder.reset_v2(contents.substr(0, dl), header.num_values);
        contents.remove_prefix(dl);
        // The reference resets the value decoder on its (stale) decompression buffer when the values are
        // not compressed (column_chunk_reader.cc:104-110); the page contents are what the format means.
        _val_decoder.reset(contents, header.encoding);
    }
    void load_dictionary_page(page p) {                             // ref: column_chunk_reader.cc:113-138
        if (!p.header->isset.dictionary_page_header) throw parquet_exception::corrupted_file("DictionaryPageHeader not set for DICTIONARY_PAGE header");
        const format::DictionaryPageHeader& header = p.header->dictionary_page_header;
        if (header.num_values < 0) throw parquet_exception::corrupted_file("Negative num_values");
        if (p.header->uncompressed_page_size < 0)
            throw parquet_exception::corrupted_file(strcat_all("Negative uncompressed_page_size in header: ", p.header->uncompressed_page_size));
        _dict = std::vector<output_type>(header.num_values);
        value_decoder<T> vd{_type_length};
        vd.reset(p.contents, format::Encoding::PLAIN);
        size_t n_read = vd.read_batch(_dict->size(), _dict->data());
        if (n_read < _dict->size())
            throw parquet_exception::corrupted_file(strcat_all("Unexpected end of dictionary page (expected ", _dict->size(), " values, got ", n_read, ")"));
        _val_decoder.reset_dict(_dict->data(), _dict->size());
    }
    void load_next_page() {                                         // ref: column_chunk_reader.cc:140-163
        ++_page_ordinal;
        std::optional<page> p = _source.next_page();
        if (!p) { _eof = true; return; }
        switch (p->header->type) {
        case format::PageType::DATA_PAGE: load_data_page(*p); _initialized = true; return;
        case format::PageType::DATA_PAGE_V2: load_data_page_v2(*p); _initialized = true; return;
        case format::PageType::DICTIONARY_PAGE: load_dictionary_page(*p); return;
        default:;   // unknown page types are skipped
        }
    }
    template <typename LevelT>
    size_t read_batch_internal(size_t n, LevelT def[], LevelT rep[], output_type val[]) {   // ref: column_chunk_reader.hh:101-146
        for (;;) {
            if (_eof) return 0;
            if (!_initialized) { load_next_page(); continue; }
            size_t def_levels_read = _def_decoder.read_batch(static_cast<uint32_t>(n), def);
            size_t rep_levels_read = _rep_decoder.read_batch(static_cast<uint32_t>(n), rep);
            if (def_levels_read != rep_levels_read)
                throw parquet_exception::corrupted_file(strcat_all("Number of definition levels ", def_levels_read,
                        " does not equal the number of repetition levels ", rep_levels_read, " in batch"));
            if (def_levels_read == 0) { _initialized = false; continue; }
            for (size_t i = 0; i < def_levels_read; ++i) {
                if (def[i] < 0 || def[i] > static_cast<LevelT>(_def_level))
                    throw parquet_exception::corrupted_file(strcat_all("Definition level (", def[i], ") out of range (0 to ", _def_level, ")"));
                if (rep[i] < 0 || rep[i] > static_cast<LevelT>(_rep_level))
                    throw parquet_exception::corrupted_file(strcat_all("Repetition level (", rep[i], ") out of range (0 to ", _rep_level, ")"));
            }
            size_t values_to_read = 0;
            for (size_t i = 0; i < def_levels_read; ++i)
                if (def[i] == static_cast<LevelT>(_def_level)) ++values_to_read;
            size_t values_read = _val_decoder.read_batch(values_to_read, val);
            if (values_read != values_to_read)
                throw parquet_exception::corrupted_file(strcat_all("Number of values in batch ", values_read,
                        " is less than indicated by def levels ", values_to_read));
            return def_levels_read;
        }
    }
public:
    explicit column_chunk_reader(page_reader&& source, format::CompressionCodec::type codec, uint32_t def_level,
                                 uint32_t rep_level, std::optional<uint32_t> type_length)
        : _source(std::move(source)), _rep_decoder(rep_level), _def_decoder(def_level), _val_decoder(type_length),
          _def_level(def_level), _rep_level(rep_level), _type_length(type_length) {
        if (codec != format::CompressionCodec::UNCOMPRESSED)
            throw parquet_exception("The GPU decode path accepts UNCOMPRESSED column chunks only");
    }
    // Read a batch of n (rep, def, value) triplets; the last batch of a page may be smaller than n.
    // Returns the number of triplets read; null values are not written to val.
    template <typename LevelT>
    size_t read_batch(size_t n, LevelT def[], LevelT rep[], output_type val[]) {   // ref: column_chunk_reader.hh:148-157
        try {
            return read_batch_internal(n, def, rep, val);
        } catch (const std::exception& e) {
            throw parquet_exception(strcat_all("Error while reading page number ", _page_ordinal, ": ", e.what()));
        }
    }
};

}  // namespace parquet4seastar_amd
