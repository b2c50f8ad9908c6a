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
//   * CompressionCodec::UNCOMPRESSED and SNAPPY are accepted (SNAPPY pages are decompressed on the GPU;
//     GZIP wraps zlib in the reference and is outside this path).
// Two ways in:
//   * value_decoder<T> / level_decoder: one page per call, synchronous - the shape of the
//     reference's unit tests;
//   * column_chunk_decoder<T>: the whole column chunk in a handful of launches (every page of
//     the chunk is one workgroup's work), results left in device memory;
//     column_chunk_reader<T> serves the reference's triplet batches out of it.
// There is no CPU decode path in this file: every level and value comes out of a kernel.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <variant>
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
    // A struct that ends without one of its `required` fields (doc/parquet/parquet.thrift:507-659): the generated
    // Thrift reader throws TProtocolException(INVALID_DATA), which the reference reports as this text
    // (include/parquet4seastar/thrift_serdes.hh:147-149).
    static void require(unsigned seen, unsigned required) {
        if ((seen & required) != required) throw parquet_exception("Could not deserialize thrift: TProtocolException: Invalid data");
    }
public:
    thrift_compact_reader(const byte* p, size_t n) : _p(p), _end(p + n), _begin(p) {}
    size_t consumed() const { return static_cast<size_t>(_p - _begin); }

    void read(format::DataPageHeader& h) {
        int16_t last = 0; int id, type;
        unsigned seen = 0;
        while (field(last, id, type)) {
            if (id >= 1 && id <= 4) seen |= 1u << id;
            switch (id) {
            case 1: h.num_values = i32(type); break;
            case 2: h.encoding = static_cast<format::Encoding::type>(i32(type)); break;
            case 3: h.definition_level_encoding = static_cast<format::Encoding::type>(i32(type)); break;
            case 4: h.repetition_level_encoding = static_cast<format::Encoding::type>(i32(type)); break;
            default: skip(type);
            }
        }
        require(seen, 0x1Eu);
    }
    void read(format::DictionaryPageHeader& h) {
        int16_t last = 0; int id, type;
        unsigned seen = 0;
        while (field(last, id, type)) {
            if (id >= 1 && id <= 2) seen |= 1u << id;
            switch (id) {
            case 1: h.num_values = i32(type); break;
            case 2: h.encoding = static_cast<format::Encoding::type>(i32(type)); break;
            case 3: h.is_sorted = (type == 1); break;
            default: skip(type);
            }
        }
        require(seen, 0x06u);
    }
    void read(format::DataPageHeaderV2& h) {
        int16_t last = 0; int id, type;
        unsigned seen = 0;
        while (field(last, id, type)) {
            if (id >= 1 && id <= 6) seen |= 1u << id;
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
        require(seen, 0x7Eu);
    }
    void read(format::PageHeader& h) {
        int16_t last = 0; int id, type;
        unsigned seen = 0;
        while (field(last, id, type)) {
            if (id >= 1 && id <= 3) seen |= 1u << id;
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
        require(seen, 0x0Eu);
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

// ref: include/parquet4seastar/encoding.hh:35-37
constexpr inline uint32_t bit_width(uint64_t max_n) {
    uint32_t w = 0;
    while (max_n) { ++w; max_n >>= 1; }
    return w;
}

// Text of the exception the reference throws for a per-page status written by a kernel.
inline std::string page_error_text(const p4s_page_status& st) {
    char buf[512];
    p4s_format_error(&st, buf, sizeof buf);
    return std::string(buf);
}

// Framing of a v1 level section (ref: level_decoder::reset_v1, src/encoding.cc:26-66): returns
// the number of bytes the section occupies and the P4S_LEVELS_* kind the kernel must be told.
// Throws what the reference throws.
// `known_len`: the section's int32 length prefix when the caller already holds it; `absent`: the bytes of `buffer`
// are not in host memory (only its size means something), so the prefix must be known.
inline size_t level_section_v1(bytes_view buffer, format::Encoding::type encoding, uint32_t num_values,
                               uint32_t width, uint32_t& kind, const int32_t* known_len = nullptr, bool absent = false) {
    kind = P4S_LEVELS_V1_RLE;
    if (width == 0) return 0;
    if (encoding == format::Encoding::RLE) {
        if (buffer.size() < 4)
            throw parquet_exception::corrupted_file(strcat_all("End of page while reading levels (needed ", 4, "B, got ", buffer.size(), "B)"));
        int32_t len;
        if (known_len) len = *known_len;
        else if (absent) throw std::logic_error("level_section_v1: the length prefix of a device-resident page was not fetched");
        else std::memcpy(&len, buffer.data(), 4);
        if (len < 0) throw parquet_exception::corrupted_file(strcat_all("Negative RLE levels length (", len, ")"));
        if (static_cast<size_t>(len) > buffer.size())
            throw parquet_exception::corrupted_file(strcat_all("End of page while reading levels (needed ", len, "B, got ", buffer.size(), "B)"));
        // the reference compares len with size() instead of size() - 4 and may run 4 bytes past the page
        return std::min<size_t>(4 + static_cast<size_t>(len), buffer.size());
    }
    if (encoding == format::Encoding::BIT_PACKED) {
        kind = P4S_LEVELS_V1_BIT_PACKED;
        const uint64_t byte_len = (static_cast<uint64_t>(num_values) * width + 7) >> 3;
        if (byte_len > 0x7FFFFFFFull)
            throw parquet_exception::corrupted_file(strcat_all("BIT_PACKED length exceeds int (", byte_len, "B)"));
        if (byte_len > buffer.size())
            throw parquet_exception::corrupted_file(strcat_all("End of page while reading levels (needed ", byte_len, "B, got ", buffer.size(), "B)"));
        return byte_len;
    }
    throw parquet_exception(strcat_all("Unknown level encoding (", encoding, ")"));
}

inline std::string out_of_memory_text(uint64_t bytes, const char* where) {
    return strcat_all("Page sizes ask for ", bytes, " bytes of ", where, " memory, which cannot be allocated");
}

// ---------------------------------------------------------------------------
// Device plumbing: one context per thread, grow-only staging buffers.
// ---------------------------------------------------------------------------
class device {
    p4s_ctx* _ctx = nullptr;
public:
    void check(int rc, const char* what) {
        if (rc != P4S_OK) throw std::runtime_error(strcat_all(what, " failed (", rc, "): ", p4s_last_error(_ctx)));
    }
    // grow-only device allocation
    class buffer {
        device* _dev = nullptr;
        void* _p = nullptr;
        uint64_t _cap = 0;
    public:
        buffer() = default;
        explicit buffer(device& d) : _dev(&d) {}
        buffer(const buffer&) = delete;
        buffer& operator=(const buffer&) = delete;
        buffer(buffer&& o) noexcept : _dev(o._dev), _p(o._p), _cap(o._cap) { o._p = nullptr; o._cap = 0; }
        buffer& operator=(buffer&& o) noexcept {
            if (this != &o) { release(); _dev = o._dev; _p = o._p; _cap = o._cap; o._p = nullptr; o._cap = 0; }
            return *this;
        }
        ~buffer() { release(); }
        void release() { if (_p) { p4s_dev_free(_dev->_ctx, _p); _p = nullptr; _cap = 0; } }
        void* ensure(uint64_t n, bool exact = false) {
            if (n > _cap || !_p) {
                release();
                const uint64_t want = exact ? std::max<uint64_t>(n, 16) : std::max<uint64_t>(n + n / 2, 4096);
                const int rc = p4s_dev_alloc(_dev->_ctx, want, &_p);
                // sizes come from page headers: a header that announces more than the device can hold is a
                // broken page, reported like every other one
                if (rc == P4S_ERR_OOM) { _p = nullptr; throw parquet_exception::corrupted_file(out_of_memory_text(want, "device")); }
                _dev->check(rc, "p4s_dev_alloc");
                _cap = want;
            }
            return _p;
        }
        void* get() const { return _p; }
        template <typename U> U* as() const { return static_cast<U*>(_p); }
        uint64_t capacity() const { return _cap; }
    };
private:
    buffer _in{*this}, _out{*this}, _aux{*this}, _desc{*this}, _status{*this}, _scratch{*this}, _dict{*this};
public:
    explicit device(int index) {
        int rc = p4s_ctx_create(index, &_ctx);
        if (rc != P4S_OK) throw std::runtime_error(strcat_all("p4s_ctx_create(", index, ") failed (", rc, "): no usable GPU"));
    }
    ~device() {
        for (buffer* b : {&_in, &_out, &_aux, &_desc, &_status, &_scratch, &_dict}) b->release();
        p4s_ctx_destroy(_ctx);
    }
    device(const device&) = delete;
    device& operator=(const device&) = delete;
    // The calling thread's device: index from P4S_DEVICE (default 0).
    static device& current() {
        static thread_local std::unique_ptr<device> d;
        if (!d) {
            const char* e = std::getenv("P4S_DEVICE");
            d = std::make_unique<device>(e ? std::atoi(e) : 0);
        }
        return *d;
    }
    p4s_ctx* ctx() { return _ctx; }
    void h2d(void* d_dst, const void* h_src, uint64_t n) { check(p4s_copy_h2d(_ctx, d_dst, h_src, n), "p4s_copy_h2d"); }
    void d2h(void* h_dst, const void* d_src, uint64_t n) { check(p4s_copy_d2h(_ctx, h_dst, d_src, n), "p4s_copy_d2h"); }
    void sync() { check(p4s_ctx_sync(_ctx), "p4s_ctx_sync"); }

    // What one page produced.
    struct result {
        p4s_page_status st{};
        [[noreturn]] void raise() const { throw parquet_exception(page_error_text(st)); }
    };

    enum class op { plain_fixed, plain_bool, bss, delta, rle, levels, dict, dlba, dba, plain_ba, flba };

    // Fixed-width output: decodes one page, copies min(values_decoded, cap) elements to `out`.
    result run_fixed(op o, bytes_view page, uint32_t n, uint32_t p0, uint32_t p1, uint32_t flags, int elem,
                     void* out, uint64_t cap, const void* h_dict = nullptr, uint64_t dict_size = 0) {
        uint8_t* d_in = static_cast<uint8_t*>(_in.ensure(page.size() + 16));
        void* d_out = _out.ensure(std::max<uint64_t>(cap, 1) * elem);
        auto* d_desc = static_cast<p4s_page_desc*>(_desc.ensure(sizeof(p4s_page_desc)));
        auto* d_st = static_cast<p4s_page_status*>(_status.ensure(sizeof(p4s_page_status)));
        p4s_page_desc pd{};
        pd.in_offset = 0; pd.in_size = page.size(); pd.out_offset = 0; pd.aux_offset = 0;
        pd.num_values = static_cast<uint32_t>(std::min<uint64_t>(n, cap));
        pd.param0 = p0; pd.param1 = p1; pd.flags = flags;
        h2d(d_in, page.data(), page.size());
        h2d(d_desc, &pd, sizeof pd);
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
            void* d_dict = _dict.ensure((dict_size + 1) * elem);
            h2d(d_dict, h_dict, (dict_size + 1) * elem);
            rc = p4s_decode_dictionary(_ctx, d_in, inb, d_desc, 1, d_dict, dict_size, elem, d_out, d_st);
            break;
        }
        default: throw std::logic_error("run_fixed: not a fixed-width decoder");
        }
        check(rc, "decode");
        result r;
        d2h(&r.st, d_st, sizeof r.st);
        sync();
        if (r.st.code == 0 && r.st.values_decoded) {
            d2h(out, d_out, std::min<uint64_t>(r.st.values_decoded, cap) * elem);
            sync();
        }
        return r;
    }

    // (offsets, bytes) output.  DELTA_BYTE_ARRAY runs the size pass first.
    result run_strings(op o, bytes_view page, uint32_t n, uint32_t flba_len, std::vector<int64_t>& offsets, std::vector<byte>& data) {
        uint8_t* d_in = static_cast<uint8_t*>(_in.ensure(page.size() + 16));
        auto* d_desc = static_cast<p4s_page_desc*>(_desc.ensure(sizeof(p4s_page_desc)));
        auto* d_st = static_cast<p4s_page_status*>(_status.ensure(sizeof(p4s_page_status)));
        const uint64_t inb = page.size();
        // a PLAIN page holds at most one string per 4 bytes, an FLBA page one per type length; asking for
        // one more than that still reaches the truncation errors of a page that ends mid-value
        uint64_t max_vals = n;
        if (o == op::plain_ba) max_vals = std::min<uint64_t>(n, inb / 4 + 1);
        if (o == op::flba && flba_len) max_vals = std::min<uint64_t>(n, inb / flba_len + 1);
        if (o == op::dlba || o == op::dba) max_vals = std::min<uint64_t>(n, delta_total(page));
        auto* d_off = static_cast<int64_t*>(_out.ensure((max_vals + 2) * 8));
        p4s_page_desc pd{};
        pd.in_size = inb; pd.num_values = static_cast<uint32_t>(max_vals);
        h2d(d_in, page.data(), page.size());
        result r;
        uint64_t cap = inb;
        if (o == op::flba) pd.param0 = flba_len;
        const uint64_t scratch_bytes = (max_vals + 2) * 8;
        if (o == op::dba) {
            void* d_scr = _scratch.ensure(scratch_bytes);
            h2d(d_desc, &pd, sizeof pd);
            check(p4s_delta_byte_array_sizes(_ctx, d_in, inb, d_desc, 1, d_scr, scratch_bytes, d_st), "dba sizes");
            d2h(&r.st, d_st, sizeof r.st);
            sync();
            if (r.st.code != 0) return r;
            cap = r.st.out_bytes;
        }
        if (o != op::flba) pd.param0 = static_cast<uint32_t>(std::min<uint64_t>(cap, 0xFFFFFFFFull));
        uint8_t* d_bytes = static_cast<uint8_t*>(_aux.ensure(cap + 16));
        h2d(d_desc, &pd, sizeof pd);
        int rc = P4S_OK;
        switch (o) {
        case op::dlba: rc = p4s_decode_delta_length_byte_array(_ctx, d_in, inb, d_desc, 1, d_off, d_bytes, cap, d_st); break;
        case op::plain_ba: rc = p4s_decode_plain_byte_array(_ctx, d_in, inb, d_desc, 1, d_off, d_bytes, cap, d_st); break;
        case op::flba: rc = p4s_decode_plain_fixed_len_byte_array(_ctx, d_in, inb, d_desc, 1, d_bytes, cap, d_st); break;
        case op::dba: rc = p4s_decode_delta_byte_array(_ctx, d_in, inb, d_desc, 1, d_off, d_bytes, cap, _scratch.get(), scratch_bytes, d_st); break;
        default: throw std::logic_error("run_strings: not a string decoder");
        }
        check(rc, "decode");
        d2h(&r.st, d_st, sizeof r.st);
        sync();
        if (r.st.code != 0) return r;
        const uint64_t nv = r.st.values_decoded, nb = r.st.out_bytes;
        offsets.assign(nv + 1, 0);
        data.resize(nb);
        if (o == op::flba) {
            for (uint64_t i = 0; i <= nv; ++i) offsets[i] = static_cast<int64_t>(i * flba_len);
        } else {
            d2h(offsets.data(), d_off, (nv + 1) * 8);
        }
        if (nb) d2h(data.data(), d_bytes, nb);
        sync();
        return r;
    }

    // Most values a DELTA_BINARY_PACKED stream of this size can yield: the count its header announces (third
    // ULEB128), but no more than the blocks the bytes can hold (a block costs at least its min_delta byte and its
    // width bytes, src/encoding.cc:157-170).  1 when the header cannot be read (the kernel reports why).
    static uint64_t delta_total(bytes_view page) {
        size_t pos = 0;
        uint64_t f[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) {
            uint64_t v = 0;
            int sh = 0;
            for (;;) {
                if (pos >= page.size() || sh > 63) return 1;
                byte b = page[pos++];
                v |= static_cast<uint64_t>(b & 0x7F) << sh;
                sh += 7;
                if (!(b & 0x80)) break;
            }
            f[k] = v;
        }
        const uint64_t block = f[0], nmb = f[1], total = std::min<uint64_t>(f[2], 0xFFFFFFFFull);
        if (nmb == 0) return 1;
        const uint64_t blocks = page.size() / (1 + std::min<uint64_t>(nmb, page.size())) + 1;
        const uint64_t room = block > 0xFFFFFFFFull / blocks ? 0xFFFFFFFFull : 1 + blocks * block;
        return std::min(total, room);
    }
};

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
    void decode() {
        _levels.assign(std::max<uint32_t>(_num_values, 1), 0);
        auto r = device::current().run_fixed(device::op::levels, _section, _num_values, _max_level, _num_values, _flags, 4,
                                             _levels.data(), _num_values);
        if (r.st.code != 0) r.raise();
        _levels.resize(r.st.values_decoded);
        _decoded = true;
    }
public:
    explicit level_decoder(uint32_t max_level) : _max_level(max_level), _bit_width(bit_width(max_level)) {}

    size_t reset_v1(bytes_view buffer, format::Encoding::type encoding, uint32_t num_values) {
        _num_values = num_values;
        _values_read = 0;
        _decoded = false;
        const size_t used = level_section_v1(buffer, encoding, num_values, _bit_width, _flags);
        _section.assign(buffer.data(), used);
        return used;
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

// Type/encoding validity (ref: value_decoder<T>::reset, src/encoding.cc:527-584).  Throws what
// the reference throws.
template <format::Type::type ParquetType>
inline void check_value_encoding(format::Encoding::type encoding, bool dict_set) {
    using T = format::Type;
    using E = format::Encoding;
    switch (encoding) {
    case E::PLAIN: return;
    case E::RLE_DICTIONARY:
    case E::PLAIN_DICTIONARY:
        if (!dict_set) throw parquet_exception::corrupted_file("No dictionary page found before a dictionary-encoded page");
        return;
    case E::RLE:
        if constexpr (ParquetType != T::BOOLEAN) throw parquet_exception::corrupted_file("RLE encoding is valid only for BOOLEAN values");
        return;
    case E::DELTA_BINARY_PACKED:
        if constexpr (ParquetType != T::INT32 && ParquetType != T::INT64)
            throw parquet_exception::corrupted_file("DELTA_BINARY_PACKED is valid only for INT32 and INT64");
        return;
    case E::DELTA_LENGTH_BYTE_ARRAY:
        if constexpr (ParquetType != T::BYTE_ARRAY) throw parquet_exception::corrupted_file("DELTA_LENGTH_BYTE_ARRAY is valid only for BYTE_ARRAY");
        return;
    case E::DELTA_BYTE_ARRAY:
        if constexpr (ParquetType != T::BYTE_ARRAY) throw parquet_exception::corrupted_file("DELTA_BYTE_ARRAY is valid only for BYTE_ARRAY");
        return;
    case E::BYTE_STREAM_SPLIT:
        if constexpr (ParquetType != T::FLOAT && ParquetType != T::DOUBLE)
            throw parquet_exception::corrupted_file("BYTE_STREAM_SPLIT is valid only for FLOAT and DOUBLE");
        return;
    default:
        throw parquet_exception(strcat_all("Encoding ", encoding, " not implemented"));
    }
}

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
        try {
            run_unchecked(n);
        } catch (const std::bad_alloc&) {
            throw parquet_exception::corrupted_file(out_of_memory_text(n * sizeof(output_type), "host"));
        } catch (const std::length_error&) {
            throw parquet_exception::corrupted_file(out_of_memory_text(n * sizeof(output_type), "host"));
        }
    }

    void run_unchecked(uint64_t n) {
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
        _page.assign(data.data(), data.size());
        _served = 0; _asked = 0; _exhausted = false;
        _values.clear();
        if (_encoding == format::Encoding::DELTA_BINARY_PACKED) {
            // the reference parses only the stream header in reset() (src/encoding.cc:178-205): a pass that asks
            // for no value reports exactly the header's errors; values are decoded when they are asked for
            run(0);
        } else if (_encoding == format::Encoding::DELTA_LENGTH_BYTE_ARRAY || _encoding == format::Encoding::DELTA_BYTE_ARRAY) {
            // the reference decodes every length in reset() (src/encoding.cc:273-293, 327-361): run the page
            // now so that what is wrong with it surfaces from reset() as well
            run(0xFFFFFFFFull);
        }
    }

    size_t read_batch(size_t n, output_type out[]) override {
        if (_served + n > _values.size() && !_exhausted && _served + n > _asked) {
            // decode ahead of the request (valid streams are batch-invariant); if the larger pass meets an error
            // that the values asked for so far do not reach, only those are decoded, as the reference would
            const uint64_t want = _served + n, ahead = std::max<uint64_t>(want, 2 * _asked);
            try {
                run(ahead);
            } catch (const parquet_exception&) {
                if (ahead == want) throw;
                run(want);
            }
        }
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
        check_value_encoding<ParquetType>(encoding, _dict_set);
        _decoder = std::make_unique<gpu_page_decoder<ParquetType>>(encoding, _type_length.value_or(0), _dict, _dict_size);
        _decoder->reset(buf);
    }
    size_t read_batch(size_t n, output_type out[]) { return _decoder->read_batch(n, out); }
};

// ---------------------------------------------------------------------------
// page_reader (ref: include/parquet4seastar/column_chunk_reader.hh:31-47,
// src/column_chunk_reader.cc:27-50) over an in-memory column chunk.
// ---------------------------------------------------------------------------
// A page whose contents lie in device memory only (the decompressed pages of a SNAPPY chunk): its header, where its
// contents are in the device image, and the two v1 level-section length prefixes (src/encoding.cc:35-50), the only
// bytes of a data page the host looks at.
struct listed_page {
    format::PageHeader header;
    uint64_t body_offset = 0;              // of the contents in the device image
    uint64_t size = 0;                     // of the contents
    int32_t v1_rep_prefix = 0, v1_def_prefix = 0;
    uint32_t known = 0;                    // 1: v1_rep_prefix was fetched, 2: v1_def_prefix was
};

struct page {
    const format::PageHeader* header;
    bytes_view contents;                   // with `listed` set only its size is meaningful
    const listed_page* listed = nullptr;
};

class page_reader {
    bytes _chunk;
    size_t _pos = 0;
    std::unique_ptr<format::PageHeader> _latest_header = std::make_unique<format::PageHeader>();
    // a page index made by p4s_walk_page_headers (the header walk on the device), served instead of parsing
    bool _indexed = false;
    std::vector<p4s_page_info> _index;
    p4s_page_status _index_status{};
    size_t _next = 0;
    // pages that lie in device memory, served as they are listed; behind the last of them `_tail_error` is thrown once
    bool _is_listed = false;
    std::vector<listed_page> _listed;
    std::string _tail_error;
public:
    explicit page_reader(bytes chunk) : _chunk(std::move(chunk)) {}
    explicit page_reader(bytes_view chunk) : _chunk(chunk) {}
    const bytes& chunk() const { return _chunk; }
    size_t position() const { return _pos; }
    // From here on next_page() serves the records of `index` (what the device found in this same chunk) and, behind
    // the last of them, throws the error that ended the device's walk, where next_page() itself would have thrown.
    void use_index(std::vector<p4s_page_info> index, const p4s_page_status& status) {
        _index = std::move(index);
        _index_status = status;
        _indexed = true;
        _next = 0;
    }
    void use_pages(std::vector<listed_page> pages, std::string tail_error) {
        _listed = std::move(pages);
        _tail_error = std::move(tail_error);
        _is_listed = true;
        _next = 0;
    }
    // The PageHeader a record of the device's page index stands for.
    static void fill_header(format::PageHeader& h, const p4s_page_info& pi) {
        h = format::PageHeader{};
        h.type = static_cast<format::PageType::type>(pi.type);
        h.uncompressed_page_size = pi.uncompressed_page_size;
        h.compressed_page_size = pi.compressed_page_size;
        if (pi.flags & P4S_PAGE_HAS_DATA_HEADER) {
            h.isset.data_page_header = true;
            h.data_page_header.num_values = pi.num_values;
            h.data_page_header.encoding = static_cast<format::Encoding::type>(pi.encoding);
            h.data_page_header.definition_level_encoding = static_cast<format::Encoding::type>(pi.definition_level_encoding);
            h.data_page_header.repetition_level_encoding = static_cast<format::Encoding::type>(pi.repetition_level_encoding);
        }
        if (pi.flags & P4S_PAGE_HAS_DICT_HEADER) {
            h.isset.dictionary_page_header = true;
            h.dictionary_page_header.num_values = pi.num_values;
            h.dictionary_page_header.encoding = static_cast<format::Encoding::type>(pi.encoding);
            h.dictionary_page_header.is_sorted = (pi.flags & P4S_PAGE_IS_SORTED) != 0;
        }
        if (pi.flags & P4S_PAGE_HAS_DATA_HEADER_V2) {
            h.isset.data_page_header_v2 = true;
            h.data_page_header_v2.num_values = pi.num_values;
            h.data_page_header_v2.num_nulls = pi.num_nulls;
            h.data_page_header_v2.num_rows = pi.num_rows;
            h.data_page_header_v2.encoding = static_cast<format::Encoding::type>(pi.encoding);
            h.data_page_header_v2.definition_levels_byte_length = pi.definition_levels_byte_length;
            h.data_page_header_v2.repetition_levels_byte_length = pi.repetition_levels_byte_length;
            h.data_page_header_v2.is_compressed = (pi.flags & P4S_PAGE_IS_COMPRESSED) != 0;
            h.data_page_header_v2.isset_is_compressed = true;
        }
    }
    // View the next page. Returns an empty result on eof.
    std::optional<page> next_page() {
        *_latest_header = format::PageHeader{};
        if (_is_listed) {
            if (_next >= _listed.size()) {
                if (!_tail_error.empty()) {
                    std::string msg = std::move(_tail_error);
                    _tail_error.clear();
                    throw parquet_exception(msg);
                }
                return std::nullopt;
            }
            const listed_page& lp = _listed[_next++];
            *_latest_header = lp.header;
            static const uint8_t nothing = 0;
            _pos = lp.body_offset + lp.size;
            return page{_latest_header.get(), bytes_view(&nothing, lp.size), &lp};
        }
        if (_indexed) {
            if (_next >= _index.size()) {
                if (_index_status.code) {
                    char msg[320];
                    p4s_format_error(&_index_status, msg, sizeof(msg));
                    _index_status.code = 0;            // thrown once, like a stream that ended in an error
                    throw parquet_exception(msg);
                }
                return std::nullopt;
            }
            const p4s_page_info& pi = _index[_next++];
            fill_header(*_latest_header, pi);
            const size_t sz = static_cast<uint32_t>(pi.compressed_page_size);
            bytes_view contents(_chunk.data() + pi.body_offset, sz);
            _pos = pi.body_offset + sz;
            return page{_latest_header.get(), contents};
        }
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

// ---------------------------------------------------------------------------
// SNAPPY column chunks.  The reference decompresses every page on the host as it loads it
// (src/column_chunk_reader.cc:68-69,104-110,131-132 -> snappy_compressor::decompress,
// src/compression.cc:49-65).  Here all pages of the chunk are decompressed by ONE launch of
// p4s_decompress_snappy and the chunk is rewritten as the UNCOMPRESSED chunk it stands for (page headers
// re-serialised with the new sizes, v2 level sections copied as they are), which the classes below decode
// as usual.  A page that libsnappy would reject keeps its place with an empty body; its message
// (compression.cc:53-63) is handed to the decoder, which reports it when the reader reaches that page.
// Pages whose headers fail the checks the reference makes before it decompresses are copied unchanged, so
// that those checks fail in the same way.
// ---------------------------------------------------------------------------
struct chunk_image {
    bytes chunk;
    std::map<int64_t, std::string> page_errors;      // page ordinal -> what loading that page throws
};

// One page of a SNAPPY chunk as the host's walk over the headers sees it.
struct snappy_item {
    format::PageHeader header;
    size_t raw_begin = 0, raw_end = 0;      // the page (header and body) as it is in the chunk
    size_t body = 0, body_size = 0;         // its body
    size_t prefix = 0;                      // leading bytes of the body that are not compressed (v2 levels)
    bool decompress = false;
    uint64_t out_off = 0;                   // of the decompressed bytes in the decompression output
    uint32_t cap = 0;
};

// The pages of `chunk` up to its end or its first broken header (from `tail_from` on nothing can be located;
// `tail_error` is what page_reader::next_page throws there).
inline std::vector<snappy_item> snappy_scan(bytes_view chunk, size_t& tail_from, std::string& tail_error) {
    using item = snappy_item;
    std::vector<item> items;
    page_reader src(chunk);
    tail_from = chunk.size();                   // from here on the chunk is copied unchanged
    for (;;) {
        const size_t begin = src.position();
        std::optional<page> p;
        try {
            p = src.next_page();
        } catch (const std::exception& ex) {
            tail_from = begin;                  // the decoder meets the same broken header at the same ordinal
            tail_error = ex.what();
            break;
        }
        if (!p) break;
        item it;
        it.header = *p->header;
        it.raw_begin = begin;
        it.raw_end = src.position();
        it.body_size = p->contents.size();
        it.body = it.raw_end - it.body_size;
        const format::PageHeader& h = it.header;
        if (h.type == format::PageType::DATA_PAGE) {
            it.decompress = h.isset.data_page_header && h.data_page_header.num_values >= 0 && h.uncompressed_page_size >= 0;
        } else if (h.type == format::PageType::DICTIONARY_PAGE) {
            it.decompress = h.isset.dictionary_page_header && h.dictionary_page_header.num_values >= 0 && h.uncompressed_page_size >= 0;
        } else if (h.type == format::PageType::DATA_PAGE_V2 && h.isset.data_page_header_v2) {
            const format::DataPageHeaderV2& v = h.data_page_header_v2;
            const bool sane = v.num_values >= 0 && v.repetition_levels_byte_length >= 0 && v.definition_levels_byte_length >= 0 &&
                              h.uncompressed_page_size >= 0;
            const uint64_t levels = sane ? static_cast<uint64_t>(v.repetition_levels_byte_length) + static_cast<uint64_t>(v.definition_levels_byte_length) : 0;
            if (sane && v.isset_is_compressed && v.is_compressed && levels <= it.body_size &&
                levels <= static_cast<uint64_t>(h.uncompressed_page_size)) {
                it.decompress = true;
                it.prefix = static_cast<size_t>(levels);
            }
        }
        if (it.decompress) it.cap = static_cast<uint32_t>(h.uncompressed_page_size) - static_cast<uint32_t>(it.prefix);
        items.push_back(std::move(it));
    }
    return items;
}

inline chunk_image snappy_chunk_image(bytes_view chunk, device& dev) {
    using item = snappy_item;
    size_t tail_from = 0;
    std::string tail_error;
    std::vector<item> items = snappy_scan(chunk, tail_from, tail_error);
    // one launch for every compressed body
    std::vector<p4s_page_desc> descs;
    uint64_t out_bytes = 0;
    for (item& it : items) {
        if (!it.decompress) continue;
        p4s_page_desc d{};
        d.in_offset = it.body + it.prefix;
        d.in_size = it.body_size - it.prefix;
        d.out_offset = out_bytes;
        d.param0 = it.cap;
        it.out_off = out_bytes;
        out_bytes += (static_cast<uint64_t>(it.cap) + 15) & ~15ull;
        descs.push_back(d);
    }
    std::vector<p4s_page_status> st(descs.size());
    bytes raw(out_bytes, 0);
    if (!descs.empty()) {
        device::buffer d_in(dev), d_desc(dev), d_st(dev), d_out(dev);
        d_in.ensure(chunk.size() + 16, true);
        d_desc.ensure(descs.size() * sizeof(p4s_page_desc), true);
        d_st.ensure(descs.size() * sizeof(p4s_page_status), true);
        d_out.ensure(out_bytes + 16, true);
        dev.h2d(d_in.get(), chunk.data(), chunk.size());
        dev.h2d(d_desc.get(), descs.data(), descs.size() * sizeof(p4s_page_desc));
        dev.check(p4s_decompress_snappy(dev.ctx(), d_in.template as<uint8_t>(), chunk.size(), d_desc.template as<p4s_page_desc>(),
                                        static_cast<uint32_t>(descs.size()), d_out.template as<uint8_t>(), out_bytes,
                                        d_st.template as<p4s_page_status>()), "p4s_decompress_snappy");
        dev.d2h(st.data(), d_st.get(), st.size() * sizeof(p4s_page_status));
        if (out_bytes) dev.d2h(raw.data(), d_out.get(), out_bytes);
        dev.sync();
    }
    chunk_image img;
    thrift_compact_writer w;
    size_t k = 0;
    int64_t ordinal = -1;
    for (const item& it : items) {
        ++ordinal;
        if (!it.decompress) {
            img.chunk.append(chunk.data() + it.raw_begin, it.raw_end - it.raw_begin);
            continue;
        }
        const p4s_page_status& ps = st[k++];
        format::PageHeader h = it.header;
        if (h.isset.data_page_header_v2) {
            h.data_page_header_v2.isset_is_compressed = true;
            h.data_page_header_v2.is_compressed = false;
        }
        if (ps.code != P4S_E_NONE) {
            img.page_errors[ordinal] = page_error_text(ps);
            h.compressed_page_size = 0;
            img.chunk.append(w.serialize(h));
            continue;
        }
        // the reference's buffer ends up as long as the Snappy stream says, which may be less than the header does
        const int32_t size = static_cast<int32_t>(it.prefix + ps.out_bytes);
        h.compressed_page_size = size;
        h.uncompressed_page_size = size;
        img.chunk.append(w.serialize(h));
        img.chunk.append(chunk.data() + it.body, it.prefix);
        img.chunk.append(raw.data() + it.out_off, ps.out_bytes);
    }
    if (tail_from < chunk.size()) img.chunk.append(chunk.data() + tail_from, chunk.size() - tail_from);
    return img;
}

// ---------------------------------------------------------------------------
// The same, without the detour through host memory: the decompressed pages stay where p4s_decompress_snappy wrote
// them.  The host keeps the page headers it has already parsed (rewritten for the uncompressed form, as above) and
// fetches the only bytes of a data page it ever looks at: the length prefixes of the v1 level sections
// (src/encoding.cc:35-50), four bytes each, gathered for all pages by one or two small launches.  Bodies that are
// stored as they are (v2 level sections, pages whose headers fail the checks made before decompression) are copied
// into the image on the device.  P4S_SNAPPY_HOST_IMAGE=1 selects the host image above instead.
// ---------------------------------------------------------------------------
struct device_chunk {
    device::buffer image;                            // page contents, each at its listed_page::body_offset
    const uint8_t* borrowed = nullptr;               // ... or the caller's device memory (16-byte aligned), when `image` is empty
    uint64_t size = 0;                               // bytes of `image` the kernels may read
    std::vector<listed_page> pages;
    std::string tail_error;                          // what reading on behind the last page throws (a broken header)
    std::map<int64_t, std::string> page_errors;      // page ordinal -> what loading that page throws
};

// The page index of a column chunk that lies in device memory (p4s_walk_page_headers, one wavefront), with the
// status that ended the walk.
inline std::vector<p4s_page_info> walk_chunk_on_device(device& dev, const uint8_t* d_chunk, uint64_t size, uint32_t max_def,
                                                       uint32_t max_rep, p4s_page_status& st) {
    // A page is never smaller than its header and a header never smaller than 8 bytes (three required integer
    // fields, a page-type struct with at least two, two stop bytes): that bounds the records.  Start small
    // and grow when the walk runs out of records.
    const uint64_t most = std::min<uint64_t>(size / 8 + 1, 0xFFFFFFFFull);
    device::buffer d_cd(dev), d_index(dev), d_st(dev);
    void* p_cd = d_cd.ensure(sizeof(p4s_chunk_desc), true);
    void* p_st = d_st.ensure(sizeof(p4s_page_status), true);
    st = p4s_page_status{};
    void* p_index = nullptr;
    for (uint64_t cap = std::min<uint64_t>(most, 4096);; cap = std::min<uint64_t>(cap * 16, most)) {
        p4s_chunk_desc cd{};
        cd.offset = 0; cd.size = size; cd.first_page = 0;
        cd.max_pages = static_cast<uint32_t>(cap);
        cd.flags = (max_rep ? uint32_t(P4S_CHUNK_HAS_REP) : 0u) | (max_def ? uint32_t(P4S_CHUNK_HAS_DEF) : 0u) |
                   uint32_t(P4S_CHUNK_UNCOMPRESSED);
        p_index = d_index.ensure(cap * sizeof(p4s_page_info), true);
        dev.h2d(p_cd, &cd, sizeof(cd));
        dev.check(p4s_walk_page_headers(dev.ctx(), d_chunk, size, static_cast<const p4s_chunk_desc*>(p_cd), 1,
                                        static_cast<p4s_page_info*>(p_index), static_cast<p4s_page_status*>(p_st)),
                  "p4s_walk_page_headers");
        dev.d2h(&st, p_st, sizeof(st));
        dev.sync();
        if (st.code != P4S_E_OUT_CAPACITY || cap >= most) break;
    }
    std::vector<p4s_page_info> index(static_cast<size_t>(st.values_decoded));
    if (!index.empty()) { dev.d2h(index.data(), p_index, index.size() * sizeof(p4s_page_info)); dev.sync(); }
    return index;
}

// An UNCOMPRESSED column chunk that already lies in device memory (read there by the storage layer, or left there
// by an earlier step): its page headers are walked on the device and no byte of it is copied to the host, except,
// when a page needs them, level-section prefixes the walk could not place (BIT_PACKED repetition levels in front of
// RLE definition levels) and string dictionary pages.  d_chunk must be 16-byte aligned and stay valid while the
// decoder made from the result is in use.
inline device_chunk device_chunk_view(const uint8_t* d_chunk, uint64_t size, device& dev, uint32_t max_def, uint32_t max_rep) {
    device_chunk dc;
    dc.image = device::buffer(dev);
    dc.borrowed = d_chunk;
    dc.size = size;
    if (size == 0) return dc;
    p4s_page_status st{};
    const std::vector<p4s_page_info> index = walk_chunk_on_device(dev, d_chunk, size, max_def, max_rep, st);
    if (st.code) dc.tail_error = page_error_text(st);
    dc.pages.resize(index.size());
    for (size_t i = 0; i < index.size(); ++i) {
        const p4s_page_info& pi = index[i];
        listed_page& lp = dc.pages[i];
        page_reader::fill_header(lp.header, pi);
        lp.body_offset = pi.body_offset;
        lp.size = static_cast<uint32_t>(pi.compressed_page_size);
        if (pi.type != 0 || !(pi.flags & P4S_PAGE_HAS_DATA_HEADER) || lp.size < 4) continue;
        // what the walk says about the v1 level sections (it reads a prefix where it can place it: include/p4s_gpu.h)
        const bool rep_rle = pi.repetition_level_encoding == static_cast<int32_t>(format::Encoding::RLE);
        const bool def_rle = pi.definition_level_encoding == static_cast<int32_t>(format::Encoding::RLE);
        if (max_rep && rep_rle) { lp.v1_rep_prefix = pi.v1_rep_prefix; lp.known |= 1u; }
        if (max_def && def_rle) {
            uint64_t at = 0;
            bool placed = true;
            if (max_rep) {
                placed = rep_rle && pi.v1_rep_prefix >= 0 && static_cast<uint64_t>(pi.v1_rep_prefix) + 4 <= lp.size;
                at = placed ? 4 + static_cast<uint64_t>(pi.v1_rep_prefix) : 0;
            }
            if (placed && lp.size - at >= 4) { lp.v1_def_prefix = pi.v1_def_prefix; lp.known |= 2u; }
        }
    }
    return dc;
}

// P4S_DEVICE_CHUNK=1: column_chunk_reader / the C wrapper put an UNCOMPRESSED chunk into device memory first and
// decode it from there through device_chunk_view, the way a caller whose chunks already lie in device memory would.
inline bool device_chunk_wanted() {
    const char* e = std::getenv("P4S_DEVICE_CHUNK");
    return e && *e && *e != '0';
}

inline bool snappy_host_image_wanted() {
    const char* e = std::getenv("P4S_SNAPPY_HOST_IMAGE");
    return e && *e && *e != '0';
}

inline device_chunk snappy_device_chunk(bytes_view chunk, device& dev, uint32_t max_def, uint32_t max_rep) {
    using item = snappy_item;
    device_chunk dc;
    dc.image = device::buffer(dev);
    size_t tail_from = 0;
    std::vector<item> items = snappy_scan(chunk, tail_from, dc.tail_error);
    // ---- layout of the image: decompressed bytes start on 16-byte boundaries (the Snappy kernel stores 16 bytes at
    // a time), the stored prefix of a v2 page right in front of them
    struct place { uint64_t contents = 0, body = 0; };
    std::vector<place> at(items.size());
    std::vector<p4s_page_desc> snappy_descs, copy_descs;
    uint64_t cur = 0;
    for (size_t i = 0; i < items.size(); ++i) {
        item& it = items[i];
        if (it.decompress) {
            at[i].body = (cur + it.prefix + 15) & ~15ull;
            at[i].contents = at[i].body - it.prefix;
            cur = at[i].body + ((static_cast<uint64_t>(it.cap) + 15) & ~15ull);
            p4s_page_desc d{};
            d.in_offset = it.body + it.prefix;
            d.in_size = it.body_size - it.prefix;
            d.out_offset = at[i].body;
            d.param0 = it.cap;
            snappy_descs.push_back(d);
            if (it.prefix) {
                p4s_page_desc c{};
                c.in_offset = it.body; c.in_size = it.prefix; c.out_offset = at[i].contents;
                c.num_values = static_cast<uint32_t>(it.prefix);
                copy_descs.push_back(c);
            }
        } else {
            at[i].contents = at[i].body = (cur + 15) & ~15ull;
            cur = at[i].contents + it.body_size;
            if (it.body_size) {
                p4s_page_desc c{};
                c.in_offset = it.body; c.in_size = it.body_size; c.out_offset = at[i].contents;
                c.num_values = static_cast<uint32_t>(it.body_size);
                copy_descs.push_back(c);
            }
        }
    }
    dc.size = cur;
    uint8_t* d_img = static_cast<uint8_t*>(dc.image.ensure(cur + 16, true));
    std::vector<p4s_page_status> st(snappy_descs.size());
    device::buffer d_in(dev), d_desc(dev), d_st(dev);
    if (!snappy_descs.empty() || !copy_descs.empty()) {
        const size_t nd = snappy_descs.size() + copy_descs.size();
        d_in.ensure(chunk.size() + 16, true);
        auto* dd = static_cast<p4s_page_desc*>(d_desc.ensure(nd * sizeof(p4s_page_desc), true));
        auto* ds = static_cast<p4s_page_status*>(d_st.ensure(nd * sizeof(p4s_page_status), true));
        dev.h2d(d_in.get(), chunk.data(), chunk.size());
        if (!snappy_descs.empty()) {
            dev.h2d(dd, snappy_descs.data(), snappy_descs.size() * sizeof(p4s_page_desc));
            dev.check(p4s_decompress_snappy(dev.ctx(), d_in.template as<uint8_t>(), chunk.size(), dd,
                                            static_cast<uint32_t>(snappy_descs.size()), d_img, cur, ds), "p4s_decompress_snappy");
            dev.d2h(st.data(), ds, st.size() * sizeof(p4s_page_status));
        }
        if (!copy_descs.empty()) {
            dev.h2d(dd + snappy_descs.size(), copy_descs.data(), copy_descs.size() * sizeof(p4s_page_desc));
            dev.check(p4s_decode_plain_fixed(dev.ctx(), d_in.template as<uint8_t>(), chunk.size(), dd + snappy_descs.size(),
                                             static_cast<uint32_t>(copy_descs.size()), d_img, ds + snappy_descs.size(), 1),
                      "p4s_decode_plain_fixed (stored page bytes)");
        }
        dev.sync();
    }
    // ---- the pages of the uncompressed chunk
    dc.pages.resize(items.size());
    size_t k = 0;
    for (size_t i = 0; i < items.size(); ++i) {
        const item& it = items[i];
        listed_page& lp = dc.pages[i];
        lp.header = it.header;
        lp.body_offset = at[i].contents;
        if (!it.decompress) { lp.size = it.body_size; continue; }
        const p4s_page_status& ps = st[k++];
        format::PageHeader& h = lp.header;
        if (h.isset.data_page_header_v2) {
            h.data_page_header_v2.isset_is_compressed = true;
            h.data_page_header_v2.is_compressed = false;
        }
        if (ps.code != P4S_E_NONE) {
            dc.page_errors[static_cast<int64_t>(i)] = page_error_text(ps);
            h.compressed_page_size = 0;
            lp.size = 0;
            continue;
        }
        // the reference's buffer ends up as long as the Snappy stream says, which may be less than the header does
        lp.size = it.prefix + ps.out_bytes;
        h.compressed_page_size = static_cast<int32_t>(lp.size);
        h.uncompressed_page_size = static_cast<int32_t>(lp.size);
    }
    // ---- the level-section prefixes of the v1 data pages: the first section's, then (behind a repetition section
    // whose length is now known) the definition section's
    const uint32_t rep_w = bit_width(max_rep), def_w = bit_width(max_def);
    auto fetch32 = [&](const std::vector<uint64_t>& offs) {
        std::vector<int32_t> vals(offs.size());
        if (offs.empty()) return vals;
        std::vector<p4s_page_desc> g(offs.size());
        for (size_t j = 0; j < offs.size(); ++j) {
            g[j] = p4s_page_desc{};
            g[j].in_offset = offs[j]; g[j].in_size = 4; g[j].out_offset = j; g[j].num_values = 1;
        }
        device::buffer d_vals(dev);
        auto* dd = static_cast<p4s_page_desc*>(d_desc.ensure(g.size() * sizeof(p4s_page_desc), true));
        auto* ds = static_cast<p4s_page_status*>(d_st.ensure(g.size() * sizeof(p4s_page_status), true));
        void* dv = d_vals.ensure(g.size() * 4, true);
        dev.h2d(dd, g.data(), g.size() * sizeof(p4s_page_desc));
        dev.check(p4s_decode_plain_fixed(dev.ctx(), d_img, cur, dd, static_cast<uint32_t>(g.size()), dv, ds, 4),
                  "p4s_decode_plain_fixed (level prefixes)");
        dev.d2h(vals.data(), dv, vals.size() * 4);
        dev.sync();
        return vals;
    };
    auto is_v1_data = [&](const listed_page& lp) {
        return lp.header.type == format::PageType::DATA_PAGE && lp.header.isset.data_page_header &&
               lp.header.data_page_header.num_values >= 0 && lp.header.uncompressed_page_size >= 0;
    };
    std::vector<uint64_t> offs;
    std::vector<size_t> who;
    for (size_t i = 0; i < dc.pages.size(); ++i) {
        const listed_page& lp = dc.pages[i];
        if (!is_v1_data(lp) || dc.page_errors.count(static_cast<int64_t>(i)) || lp.size < 4) continue;
        const format::DataPageHeader& h = lp.header.data_page_header;
        const bool first_is_rep = rep_w != 0;
        if (!first_is_rep && def_w == 0) continue;
        if ((first_is_rep ? h.repetition_level_encoding : h.definition_level_encoding) != format::Encoding::RLE) continue;
        offs.push_back(lp.body_offset);
        who.push_back(i);
    }
    std::vector<int32_t> got = fetch32(offs);
    for (size_t j = 0; j < who.size(); ++j) {
        listed_page& lp = dc.pages[who[j]];
        if (rep_w) { lp.v1_rep_prefix = got[j]; lp.known |= 1u; } else { lp.v1_def_prefix = got[j]; lp.known |= 2u; }
    }
    if (rep_w && def_w) {
        offs.clear(); who.clear();
        static const uint8_t nothing = 0;
        for (size_t i = 0; i < dc.pages.size(); ++i) {
            const listed_page& lp = dc.pages[i];
            if (!is_v1_data(lp) || dc.page_errors.count(static_cast<int64_t>(i))) continue;
            const format::DataPageHeader& h = lp.header.data_page_header;
            if (h.definition_level_encoding != format::Encoding::RLE) continue;
            size_t rep_len = 0;
            uint32_t kind = 0;
            try {
                rep_len = level_section_v1(bytes_view(&nothing, lp.size), h.repetition_level_encoding,
                                           static_cast<uint32_t>(h.num_values), rep_w, kind,
                                           (lp.known & 1u) ? &lp.v1_rep_prefix : nullptr, true);
            } catch (const parquet_exception&) {
                continue;                   // loading the page fails at its repetition levels, before the definition levels
            }
            if (lp.size - rep_len < 4) continue;
            offs.push_back(lp.body_offset + rep_len);
            who.push_back(i);
        }
        got = fetch32(offs);
        for (size_t j = 0; j < who.size(); ++j) { dc.pages[who[j]].v1_def_prefix = got[j]; dc.pages[who[j]].known |= 2u; }
    }
    return dc;
}

// ---------------------------------------------------------------------------
// column_chunk_decoder<T>: every page of a column chunk decoded by a handful of launches.
//
//   plan()    host: walks the page headers (Thrift), splits every data page into its
//             repetition-level / definition-level / value sections exactly as
//             load_data_page / load_data_page_v2 do (ref: src/column_chunk_reader.cc:52-111),
//             uploads the chunk and three page-descriptor arrays;
//   launch()  device: dictionary pages -> dictionary buffers, rep levels, def levels,
//             values_to_read per page (count of def == max_def, ref column_chunk_reader.hh:134-139)
//             patched into the value descriptors on the device, then one launch per
//             (encoding, dictionary) group of value pages.  Nothing is copied back;
//   collect() one synchronisation, per-page statuses -> counts and the reference's error texts.
//
// Output layout in device memory: page p owns level slots [first, first + num_values) of the
// def / rep arrays (int32) and the same slots of the value array; its values_decoded values
// are dense at the start of its slot range.  String pages own offsets slots
// [first + data_page_index, ... + num_values] and a byte range of the bytes buffer.
// ---------------------------------------------------------------------------
template <format::Type::type T>
class column_chunk_decoder {
public:
    using output_type = typename value_decoder_traits<T>::output_type;
    static constexpr bool is_string = T == format::Type::BYTE_ARRAY || T == format::Type::FIXED_LEN_BYTE_ARRAY;
    // what a value slot of the fixed-width value array holds: the value, or for string types
    // (dictionary pages only) the uint32 dictionary index
    static constexpr int slot_elem = is_string ? 4 : static_cast<int>(sizeof(output_type));

    enum class value_kind : int { plain_fixed, plain_bool, rle_bool, bss, delta, dict, plain_ba, flba, dlba, dba };

    struct page_entry {
        enum kind_t { data, dictionary, skipped, broken };
        int64_t ordinal = 0;
        kind_t kind = skipped;
        std::string error;          // what the reference would throw while loading / reading this page
        uint32_t num_values = 0;    // page header
        uint64_t first = 0;         // first level slot == first value slot
        uint32_t n_def = 0, n_rep = 0;          // levels a read_batch can get
        uint32_t above_def = 0, above_rep = 0;  // levels above the maximum (rejected by the reader)
        uint32_t n_val = 0;         // values decoded
        value_kind vkind = value_kind::plain_fixed;
        int dict = -1;              // which dictionary of the chunk
        int launch_index = -1;      // position in the descriptor / status arrays
        uint64_t str_slot = 0;      // first offsets slot (string decoders)
        uint64_t bytes_off = 0, bytes_len = 0;  // range of the bytes buffer
    };

    struct dictionary {
        uint64_t size = 0;              // values
        uint64_t dev_off = 0;           // byte offset in the dictionary buffer
        std::vector<output_type> host;  // string types: the strings (the device holds the identity table)
    };

private:
    device& _dev;
    bytes_view _chunk;
    uint64_t _chunk_bytes = 0;                 // size of the chunk as the kernels read it (_d_chunk)
    bool _on_device = false;                   // the page contents were put into _d_chunk by the caller: `_listed` names them
    const uint8_t* _d_borrowed = nullptr;      // ... or lie in the caller's own device memory
    const uint8_t* d_chunk_ptr() const { return _d_borrowed ? _d_borrowed : _d_chunk.template as<uint8_t>(); }
    std::vector<listed_page> _listed;
    std::string _listed_tail_error;
    uint32_t _max_def, _max_rep;
    std::optional<uint32_t> _type_length;
    std::vector<page_entry> _pages;
    std::vector<dictionary> _dicts;
    std::vector<size_t> _order;     // launch position -> index into _pages
    struct group { value_kind kind; int dict; uint32_t begin, count; };
    std::vector<group> _groups;
    struct dict_copy { uint64_t in_off, in_size, dev_off, count; };
    std::vector<dict_copy> _dict_copies;
    std::vector<p4s_page_desc> _rep_descs, _def_descs, _val_descs;
    std::vector<p4s_page_status> _rep_st, _def_st, _val_st;
    uint64_t _total_slots = 0, _bytes_cap = 0, _dict_bytes = 0;
    bool _planned = false, _launched = false, _collected = false, _positional_done = false;
    bool _device_walk = [] { const char* e = std::getenv("P4S_DEVICE_HEADER_WALK"); return e && *e && *e != '0'; }();
    device::buffer _d_chunk, _d_descs, _d_status, _d_rep, _d_def, _d_values, _d_offsets, _d_bytes, _d_scratch, _d_dict, _d_dict_descs;
    device::buffer _d_positional, _d_validity;          // positional()
    std::vector<uint32_t> _ident;
    std::map<int64_t, std::string> _preset_errors;      // pages that failed before this decoder saw them (decompression)

    static constexpr int dict_elem() { return slot_elem; }

    void fail_page(page_entry& e, const std::exception& ex) { e.error = ex.what(); }

    // value section -> kind (ref: value_decoder<T>::reset dispatch, src/encoding.cc:527-584)
    value_kind kind_of(format::Encoding::type enc) const {
        using E = format::Encoding;
        switch (enc) {
        case E::PLAIN:
            if constexpr (T == format::Type::BOOLEAN) return value_kind::plain_bool;
            else if constexpr (T == format::Type::BYTE_ARRAY) return value_kind::plain_ba;
            else if constexpr (T == format::Type::FIXED_LEN_BYTE_ARRAY) return value_kind::flba;
            else return value_kind::plain_fixed;
        case E::RLE: return value_kind::rle_bool;
        case E::DELTA_BINARY_PACKED: return value_kind::delta;
        case E::DELTA_LENGTH_BYTE_ARRAY: return value_kind::dlba;
        case E::DELTA_BYTE_ARRAY: return value_kind::dba;
        case E::BYTE_STREAM_SPLIT: return value_kind::bss;
        default: return value_kind::dict;
        }
    }

    // `base` = offset of the page's contents inside the chunk
    void load_dictionary(page_entry& e, const page& p, uint64_t base) {
        if (!p.header->isset.dictionary_page_header)
            throw parquet_exception::corrupted_file("DictionaryPageHeader not set for DICTIONARY_PAGE header");
        const format::DictionaryPageHeader& header = p.header->dictionary_page_header;
        if (header.num_values < 0) throw parquet_exception::corrupted_file("Negative num_values");
        if (p.header->uncompressed_page_size < 0)
            throw parquet_exception::corrupted_file(strcat_all("Negative uncompressed_page_size in header: ", p.header->uncompressed_page_size));
        if (static_cast<uint64_t>(p.header->uncompressed_page_size) < p.contents.size())      // column_chunk_reader.cc:128-129, compression.cc:31-33
            throw parquet_exception::corrupted_file("Uncompression buffer size too small");
        dictionary d;
        d.size = static_cast<uint64_t>(header.num_values);
        if constexpr (is_string) {
            // the strings are decoded now (PLAIN kernels); the device gets the identity table so that the
            // dictionary kernel emits bounds-checked indices
            d.host.resize(d.size);
            value_decoder<T> vd{_type_length};
            bytes fetched;
            bytes_view body = p.contents;
            if (p.listed) {                 // the page lies in device memory: this decode wants it on the host
                fetched.resize(p.contents.size());
                if (!fetched.empty()) { _dev.d2h(fetched.data(), d_chunk_ptr() + base, fetched.size()); _dev.sync(); }
                body = fetched;
            }
            vd.reset(body, format::Encoding::PLAIN);
            const size_t n_read = vd.read_batch(d.size, d.host.data());
            if (n_read < d.size)
                throw parquet_exception::corrupted_file(strcat_all("Unexpected end of dictionary page (expected ", d.size, " values, got ", n_read, ")"));
        } else {
            const uint64_t have = p.contents.size() / sizeof(output_type);      // plain_decoder_trivial, ref src/encoding.cc:415-424
            if (have < d.size)
                throw parquet_exception::corrupted_file(strcat_all("Unexpected end of dictionary page (expected ", d.size, " values, got ", have, ")"));
            _dict_copies.push_back({base, p.contents.size(), 0, d.size});
        }
        d.dev_off = _dict_bytes;
        if (!is_string) _dict_copies.back().dev_off = d.dev_off;
        // one spare element: index == size is accepted (ref :489).  Every dictionary starts at a multiple of 48
        // bytes: 16-byte aligned and a whole number of elements of every width (4, 8, 12) from the buffer's start,
        // which is how the copy kernel addresses it (element offsets).
        _dict_bytes += ((d.size + 1) * dict_elem() + 47) / 48 * 48;
        _dicts.push_back(std::move(d));
        e.dict = static_cast<int>(_dicts.size()) - 1;
    }

    void load_data(page_entry& e, const page& p, uint64_t base, int current_dict) {
        using E = format::Encoding;
        const bool v2 = p.header->type == format::PageType::DATA_PAGE_V2;
        E::type encoding;
        int32_t num_values;
        bytes_view contents = p.contents;
        uint64_t rep_off = 0, rep_len = 0, def_off = 0, def_len = 0;
        uint32_t rep_kind = P4S_LEVELS_V2, def_kind = P4S_LEVELS_V2;
        const uint32_t rep_w = bit_width(_max_rep), def_w = bit_width(_max_def);
        if (!v2) {                                                     // ref: load_data_page, column_chunk_reader.cc:52-78
            if (!p.header->isset.data_page_header)
                throw parquet_exception::corrupted_file("DataPageHeader not set for DATA_PAGE header");
            const format::DataPageHeader& header = p.header->data_page_header;
            if (header.num_values < 0) throw parquet_exception::corrupted_file(strcat_all("Negative num_values in header: ", header.num_values));
            if (p.header->uncompressed_page_size < 0)
                throw parquet_exception::corrupted_file(strcat_all("Negative uncompressed_page_size in header: ", p.header->uncompressed_page_size));
            // the whole page goes through uncompressed_compressor::decompress into a buffer of
            // uncompressed_page_size bytes (column_chunk_reader.cc:68-69, compression.cc:29-37)
            if (static_cast<uint64_t>(p.header->uncompressed_page_size) < contents.size())
                throw parquet_exception::corrupted_file("Uncompression buffer size too small");
            num_values = header.num_values;
            encoding = header.encoding;
            // (a page that lies in device memory only: its two length prefixes were fetched when it was listed)
            const bool absent = p.listed != nullptr;
            const int32_t* rep_prefix = absent && (p.listed->known & 1u) ? &p.listed->v1_rep_prefix : nullptr;
            const int32_t* def_prefix = absent && (p.listed->known & 2u) ? &p.listed->v1_def_prefix : nullptr;
            // a prefix nobody fetched yet (the device's walk cannot place a section behind BIT_PACKED levels): four bytes now
            int32_t fetched_rep = 0, fetched_def = 0;
            auto fetch = [&](uint64_t at, int32_t& v) { _dev.d2h(&v, d_chunk_ptr() + base + at, 4); _dev.sync(); return &v; };
            if (absent && !rep_prefix && rep_w && header.repetition_level_encoding == E::RLE && contents.size() >= 4)
                rep_prefix = fetch(0, fetched_rep);
            rep_len = level_section_v1(contents, header.repetition_level_encoding, num_values, rep_w, rep_kind, rep_prefix, absent);
            contents.remove_prefix(rep_len);
            def_off = rep_len;
            if (absent && !def_prefix && def_w && header.definition_level_encoding == E::RLE && contents.size() >= 4)
                def_prefix = fetch(rep_len, fetched_def);
            def_len = level_section_v1(contents, header.definition_level_encoding, num_values, def_w, def_kind, def_prefix, absent);
            contents.remove_prefix(def_len);
        } else {                                                       // ref: load_data_page_v2, column_chunk_reader.cc:80-111
            if (!p.header->isset.data_page_header_v2)
                throw parquet_exception::corrupted_file("DataPageHeaderV2 not set for DATA_PAGE_V2 header");
            const format::DataPageHeaderV2& header = p.header->data_page_header_v2;
            if (header.num_values < 0) throw parquet_exception::corrupted_file(strcat_all("Negative num_values in header: ", header.num_values));
            if (header.repetition_levels_byte_length < 0 || header.definition_levels_byte_length < 0)
                throw parquet_exception::corrupted_file("Negative levels byte length in header");
            if (p.header->uncompressed_page_size < 0)
                throw parquet_exception::corrupted_file(strcat_all("Negative uncompressed_page_size in header: ", p.header->uncompressed_page_size));
            num_values = header.num_values;
            encoding = header.encoding;
            rep_len = static_cast<uint64_t>(header.repetition_levels_byte_length);
            def_len = static_cast<uint64_t>(header.definition_levels_byte_length);
            // the reference's substr/remove_prefix pair is undefined when the lengths exceed the page
            if (rep_len + def_len > contents.size()) throw parquet_exception::corrupted_file("Levels byte length exceeds the page");
            if (rep_len > 0x7FFFFFFFull) throw parquet_exception::corrupted_file(strcat_all("Levels length exceeds int (", rep_len, "B)"));
            if (def_len > 0x7FFFFFFFull) throw parquet_exception::corrupted_file(strcat_all("Levels length exceeds int (", def_len, "B)"));
            def_off = rep_len;
            contents.remove_prefix(rep_len + def_len);
            // The reference resets the value decoder on its (stale) decompression buffer when the values are
            // not compressed (column_chunk_reader.cc:104-110); the page contents are what the format means.
            // A page that says is_compressed in a chunk whose codec is UNCOMPRESSED goes through the reference's
            // uncompressed_compressor: a copy into a buffer of uncompressed_page_size - levels bytes
            // (column_chunk_reader.cc:104-108, compression.cc:29-37).
            if (header.isset_is_compressed && header.is_compressed) {
                const uint64_t ups = static_cast<uint64_t>(p.header->uncompressed_page_size);
                if (ups < rep_len + def_len || ups - (rep_len + def_len) < contents.size())
                    throw parquet_exception::corrupted_file("Uncompression buffer size too small");
            }
        }
        check_value_encoding<T>(encoding, current_dict >= 0);
        e.num_values = static_cast<uint32_t>(num_values);
        e.vkind = kind_of(encoding);
        e.dict = e.vkind == value_kind::dict ? current_dict : -1;
        e.kind = page_entry::data;

        p4s_page_desc rd{}, dd{}, vd{};
        rd.in_offset = base + rep_off; rd.in_size = rep_len; rd.num_values = e.num_values;
        rd.param0 = _max_rep; rd.param1 = e.num_values; rd.flags = rep_kind;
        dd.in_offset = base + def_off; dd.in_size = def_len; dd.num_values = e.num_values;
        dd.param0 = _max_def; dd.param1 = e.num_values; dd.flags = def_kind;
        vd.in_offset = base + def_off + def_len; vd.in_size = contents.size(); vd.num_values = e.num_values;
        if (e.vkind == value_kind::rle_bool) vd.param0 = 1;
        if (e.vkind == value_kind::flba) vd.param0 = _type_length.value_or(0);
        _tmp_descs.push_back({rd, dd, vd});
    }
    struct desc3 { p4s_page_desc rep, def, val; };
    std::vector<desc3> _tmp_descs;      // data pages in chunk order, before grouping

public:
    column_chunk_decoder(bytes_view chunk, uint32_t def_level, uint32_t rep_level, std::optional<uint32_t> type_length,
                         device& dev = device::current())
        : _dev(dev), _chunk(chunk), _chunk_bytes(chunk.size()), _max_def(def_level), _max_rep(rep_level), _type_length(type_length),
          _d_chunk(dev), _d_descs(dev), _d_status(dev), _d_rep(dev), _d_def(dev), _d_values(dev), _d_offsets(dev),
          _d_bytes(dev), _d_scratch(dev), _d_dict(dev), _d_dict_descs(dev), _d_positional(dev), _d_validity(dev) {
        if constexpr (T == format::Type::FIXED_LEN_BYTE_ARRAY) {
            if (!_type_length) throw parquet_exception::corrupted_file("type_length not set for FIXED_LEN_BYTE_ARRAY");
        }
    }

    // Over a chunk whose page contents already lie in device memory (snappy_device_chunk): nothing is uploaded.
    column_chunk_decoder(device_chunk&& dc, uint32_t def_level, uint32_t rep_level, std::optional<uint32_t> type_length,
                         device& dev = device::current())
        : column_chunk_decoder(bytes_view(), def_level, rep_level, type_length, dev) {
        _d_chunk = std::move(dc.image);
        _d_borrowed = dc.borrowed;
        _chunk_bytes = dc.size;
        _listed = std::move(dc.pages);
        _listed_tail_error = std::move(dc.tail_error);
        _on_device = true;
        _preset_errors = std::move(dc.page_errors);
    }

    // Pages (by ordinal) whose loading already failed, with the reference's message: see snappy_chunk_image.
    void set_page_errors(std::map<int64_t, std::string> errors) { _preset_errors = std::move(errors); }
    const std::vector<page_entry>& pages() const { return _pages; }
    const std::vector<dictionary>& dictionaries() const { return _dicts; }
    uint64_t total_slots() const { return _total_slots; }
    uint32_t max_def() const { return _max_def; }
    uint32_t max_rep() const { return _max_rep; }
    // device results (valid after launch(); contents complete after collect())
    const int32_t* d_def() const { return _d_def.template as<int32_t>(); }
    const int32_t* d_rep() const { return _d_rep.template as<int32_t>(); }
    const void* d_values() const { return _d_values.get(); }
    const int64_t* d_offsets() const { return _d_offsets.template as<int64_t>(); }
    const uint8_t* d_bytes() const { return _d_bytes.template as<uint8_t>(); }

    // The page-header walk runs on the device (p4s_walk_page_headers over the uploaded chunk) instead of on the host.
    // Default: the environment variable P4S_DEVICE_HEADER_WALK (non-zero).
    void set_device_header_walk(bool on) { _device_walk = on; }

    void plan() {
        if (_planned) return;
        page_reader src(_chunk);
        if (_on_device) {
            src.use_pages(std::move(_listed), std::move(_listed_tail_error));
        } else if (_device_walk && !_chunk.empty()) {
            uint8_t* d_chunk = static_cast<uint8_t*>(_d_chunk.ensure(_chunk.size() + 16, true));
            _dev.h2d(d_chunk, _chunk.data(), _chunk.size());
            p4s_page_status st{};
            std::vector<p4s_page_info> index = walk_chunk_on_device(_dev, d_chunk, _chunk.size(), _max_def, _max_rep, st);
            src.use_index(std::move(index), st);
        }
        int current_dict = -1;
        int64_t ordinal = -1;
        std::vector<size_t> data_pages;
        for (;;) {
            ++ordinal;
            page_entry e;
            e.ordinal = ordinal;
            std::optional<page> p;
            try {
                p = src.next_page();
            } catch (const std::exception& ex) {
                // the stream position is lost: nothing after this page can be located
                e.kind = page_entry::broken;
                fail_page(e, ex);
                _pages.push_back(std::move(e));
                break;
            }
            if (!p) break;
            if (auto pe = _preset_errors.find(ordinal); pe != _preset_errors.end()) {
                e.kind = p->header->type == format::PageType::DICTIONARY_PAGE ? page_entry::dictionary : page_entry::data;
                e.error = pe->second;
                _pages.push_back(std::move(e));
                continue;
            }
            try {
                switch (p->header->type) {
                case format::PageType::DATA_PAGE:
                case format::PageType::DATA_PAGE_V2:
                    e.kind = page_entry::data;     // a data page that fails to load is still a (failed) data page
                    load_data(e, *p, src.position() - p->contents.size(), current_dict);
                    break;
                case format::PageType::DICTIONARY_PAGE:
                    e.kind = page_entry::dictionary;
                    load_dictionary(e, *p, src.position() - p->contents.size());
                    current_dict = e.dict;
                    break;
                default: e.kind = page_entry::skipped;       // unknown page types are skipped (ref :159)
                }
            } catch (const std::exception& ex) {
                fail_page(e, ex);
            }
            if (e.kind == page_entry::data && e.error.empty()) data_pages.push_back(_pages.size());
            _pages.push_back(std::move(e));
        }
        // slots, in chunk order
        uint64_t slot = 0, k = 0;
        for (size_t pi : data_pages) {
            page_entry& e = _pages[pi];
            e.first = slot;
            e.str_slot = slot + k;
            slot += e.num_values;
            ++k;
        }
        _total_slots = slot;
        // launch order: grouped by (kind, dictionary), chunk order inside a group
        std::vector<size_t> pos(data_pages.size());
        for (size_t i = 0; i < pos.size(); ++i) pos[i] = i;
        std::stable_sort(pos.begin(), pos.end(), [&](size_t a, size_t b) {
            const page_entry &x = _pages[data_pages[a]], &y = _pages[data_pages[b]];
            if (x.vkind != y.vkind) return static_cast<int>(x.vkind) < static_cast<int>(y.vkind);
            return x.dict < y.dict;
        });
        const size_t n = pos.size();
        _order.resize(n);
        _rep_descs.resize(n); _def_descs.resize(n); _val_descs.resize(n);
        uint64_t bytes_cap = 0;
        for (size_t li = 0; li < n; ++li) {
            const size_t di = pos[li];
            page_entry& e = _pages[data_pages[di]];
            e.launch_index = static_cast<int>(li);
            _order[li] = data_pages[di];
            desc3 d = _tmp_descs[di];
            d.rep.out_offset = d.def.out_offset = e.first;
            const bool strings = e.vkind == value_kind::plain_ba || e.vkind == value_kind::flba ||
                                 e.vkind == value_kind::dlba || e.vkind == value_kind::dba;
            d.val.out_offset = strings ? e.str_slot : e.first;
            if (strings && e.vkind != value_kind::dba) {
                // PLAIN, FLBA and DELTA_LENGTH output never exceeds the encoded section
                e.bytes_off = bytes_cap;
                d.val.aux_offset = bytes_cap;
                if (e.vkind != value_kind::flba) d.val.param0 = static_cast<uint32_t>(std::min<uint64_t>(d.val.in_size, 0xFFFFFFFFull));
                bytes_cap += (d.val.in_size + 15) & ~15ull;
            }
            _rep_descs[li] = d.rep; _def_descs[li] = d.def; _val_descs[li] = d.val;
            if (_groups.empty() || _groups.back().kind != e.vkind || _groups.back().dict != e.dict)
                _groups.push_back({e.vkind, e.dict, static_cast<uint32_t>(li), 0});
            ++_groups.back().count;
        }
        _tmp_descs.clear();
        _bytes_cap = bytes_cap;
        _static_bytes_cap = bytes_cap;
        _rep_st.assign(n, p4s_page_status{}); _def_st.assign(n, p4s_page_status{}); _val_st.assign(n, p4s_page_status{});

        // ---- device side of the plan: the chunk and the descriptors
        if (!_on_device) {
            uint8_t* d_chunk = static_cast<uint8_t*>(_d_chunk.ensure(_chunk.size() + 16, true));
            if (!_chunk.empty()) _dev.h2d(d_chunk, _chunk.data(), _chunk.size());
        }
        if (n) {
            auto* dd = static_cast<p4s_page_desc*>(_d_descs.ensure(3 * n * sizeof(p4s_page_desc), true));
            _dev.h2d(dd, _rep_descs.data(), n * sizeof(p4s_page_desc));
            _dev.h2d(dd + n, _def_descs.data(), n * sizeof(p4s_page_desc));
            _dev.h2d(dd + 2 * n, _val_descs.data(), n * sizeof(p4s_page_desc));
            _d_status.ensure(3 * n * sizeof(p4s_page_status), true);
            if (_max_rep) _d_rep.ensure(std::max<uint64_t>(_total_slots, 1) * 4, true);
            if (_max_def) _d_def.ensure(std::max<uint64_t>(_total_slots, 1) * 4, true);
            bool any_fixed = false, any_str = false, any_dba = false;
            for (const group& g : _groups) {
                const bool s = g.kind == value_kind::plain_ba || g.kind == value_kind::flba || g.kind == value_kind::dlba || g.kind == value_kind::dba;
                any_str |= s; any_fixed |= !s; any_dba |= g.kind == value_kind::dba;
            }
            if (any_fixed) _d_values.ensure(std::max<uint64_t>(_total_slots, 1) * slot_elem, true);
            if (any_str) {
                _d_offsets.ensure((_total_slots + n + 1) * 8, true);
                if (_bytes_cap) _d_bytes.ensure(_bytes_cap + 16, true);
            }
            if (any_dba) _d_scratch.ensure((_total_slots + n + 1) * 8, true);
        }
        if (_dict_bytes) {
            uint8_t* dd = static_cast<uint8_t*>(_d_dict.ensure(_dict_bytes, true));
            _dev.check(p4s_memset(_dev.ctx(), dd, 0, _dict_bytes), "p4s_memset");
            if constexpr (is_string) {
                for (const dictionary& d : _dicts) {
                    _ident.resize(d.size + 1);
                    for (uint64_t i = 0; i <= d.size; ++i) _ident[i] = static_cast<uint32_t>(i);
                    _dev.h2d(dd + d.dev_off, _ident.data(), (d.size + 1) * 4);
                    _dev.sync();      // _ident is reused
                }
            } else if (!_dict_copies.empty()) {
                std::vector<p4s_page_desc> dc(_dict_copies.size());
                for (size_t i = 0; i < dc.size(); ++i) {
                    dc[i] = p4s_page_desc{};
                    dc[i].in_offset = _dict_copies[i].in_off;
                    dc[i].in_size = _dict_copies[i].in_size;
                    dc[i].out_offset = _dict_copies[i].dev_off / dict_elem();
                    dc[i].num_values = static_cast<uint32_t>(_dict_copies[i].count);
                }
                void* ddc = _d_dict_descs.ensure(dc.size() * (sizeof(p4s_page_desc) + sizeof(p4s_page_status)), true);
                _dev.h2d(ddc, dc.data(), dc.size() * sizeof(p4s_page_desc));
                _dev.sync();          // dc goes out of scope
            }
        }
        _dev.sync();
        _planned = true;
    }

    // Enqueues every kernel of the chunk; returns without waiting.  May be called again: it
    // recomputes the same outputs (a chunk with DELTA_BYTE_ARRAY pages synchronises once inside, to size their output).
    void launch() {
        plan();
        p4s_ctx* c = _dev.ctx();
        const uint32_t n = static_cast<uint32_t>(_order.size());
        const uint8_t* d_chunk = d_chunk_ptr();
        const uint64_t in_bytes = _chunk_bytes;
        if (!is_string && !_dict_copies.empty()) {
            auto* ddc = _d_dict_descs.template as<p4s_page_desc>();
            auto* dst = reinterpret_cast<p4s_page_status*>(ddc + _dict_copies.size());
            _dev.check(p4s_decode_plain_fixed(c, d_chunk, in_bytes, ddc, static_cast<uint32_t>(_dict_copies.size()),
                                              _d_dict.get(), dst, dict_elem()), "dictionary pages");
        }
        if (n == 0) { _launched = true; return; }
        auto* d_rep_desc = _d_descs.template as<p4s_page_desc>();
        auto* d_def_desc = d_rep_desc + n;
        auto* d_val_desc = d_rep_desc + 2 * n;
        auto* d_rep_st = _d_status.template as<p4s_page_status>();
        auto* d_def_st = d_rep_st + n;
        auto* d_val_st = d_rep_st + 2 * n;
        if (_max_rep)
            _dev.check(p4s_decode_levels(c, d_chunk, in_bytes, d_rep_desc, n, _d_rep.get(), d_rep_st, 4), "repetition levels");
        if (_max_def) {
            _dev.check(p4s_decode_levels(c, d_chunk, in_bytes, d_def_desc, n, _d_def.get(), d_def_st, 4), "definition levels");
            _dev.check(p4s_pages_set_num_values_from_levels(c, d_val_desc, d_def_st, n), "values_to_read");
        }
        // DELTA_BYTE_ARRAY groups go first: sizing their output may replace the bytes buffer, which
        // must happen before the other BYTE_ARRAY groups write into it
        for (const group& g : _groups)
            if (g.kind == value_kind::dba) _dev.check(launch_dba(g, d_chunk, in_bytes, d_val_desc, d_val_st), "DELTA_BYTE_ARRAY pages");
        for (const group& g : _groups) {
            const p4s_page_desc* pd = d_val_desc + g.begin;
            p4s_page_status* st = d_val_st + g.begin;
            int rc = P4S_OK;
            switch (g.kind) {
            case value_kind::plain_fixed: rc = p4s_decode_plain_fixed(c, d_chunk, in_bytes, pd, g.count, _d_values.get(), st, slot_elem); break;
            case value_kind::plain_bool: rc = p4s_decode_plain_boolean(c, d_chunk, in_bytes, pd, g.count, _d_values.template as<uint8_t>(), st); break;
            case value_kind::rle_bool: rc = p4s_decode_rle(c, d_chunk, in_bytes, pd, g.count, _d_values.get(), st, 1); break;
            case value_kind::bss: rc = p4s_decode_byte_stream_split(c, d_chunk, in_bytes, pd, g.count, _d_values.get(), st, slot_elem); break;
            case value_kind::delta: rc = p4s_decode_delta_binary_packed(c, d_chunk, in_bytes, pd, g.count, _d_values.get(), st, slot_elem); break;
            case value_kind::dict:
                rc = p4s_decode_dictionary(c, d_chunk, in_bytes, pd, g.count, _d_dict.template as<uint8_t>() + _dicts[g.dict].dev_off,
                                           _dicts[g.dict].size, dict_elem(), _d_values.get(), st);
                break;
            case value_kind::plain_ba:
                rc = p4s_decode_plain_byte_array(c, d_chunk, in_bytes, pd, g.count, _d_offsets.template as<int64_t>(),
                                                 _d_bytes.template as<uint8_t>(), _bytes_cap, st);
                break;
            case value_kind::flba:
                rc = p4s_decode_plain_fixed_len_byte_array(c, d_chunk, in_bytes, pd, g.count, _d_bytes.template as<uint8_t>(), _bytes_cap, st);
                break;
            case value_kind::dlba:
                rc = p4s_decode_delta_length_byte_array(c, d_chunk, in_bytes, pd, g.count, _d_offsets.template as<int64_t>(),
                                                        _d_bytes.template as<uint8_t>(), _bytes_cap, st);
                break;
            case value_kind::dba: break;
            }
            _dev.check(rc, "value pages");
        }
        _launched = true;
        _collected = false;
    }

private:
    // DELTA_BYTE_ARRAY output is not bounded by the encoded size: a sizes pass, one
    // synchronisation to lay the byte ranges out, then the decode.
    int launch_dba(const group& g, const uint8_t* d_chunk, uint64_t in_bytes, p4s_page_desc* d_val_desc, p4s_page_status* d_val_st) {
        p4s_ctx* c = _dev.ctx();
        const uint64_t scratch_bytes = _d_scratch.capacity();
        int rc = p4s_delta_byte_array_sizes(c, d_chunk, in_bytes, d_val_desc + g.begin, g.count, _d_scratch.get(), scratch_bytes,
                                            d_val_st + g.begin);
        if (rc != P4S_OK) return rc;
        std::vector<p4s_page_status> st(g.count);
        std::vector<p4s_page_desc> pd(g.count);
        _dev.d2h(st.data(), d_val_st + g.begin, g.count * sizeof(p4s_page_status));
        _dev.d2h(pd.data(), d_val_desc + g.begin, g.count * sizeof(p4s_page_desc));     // num_values as patched on the device
        _dev.sync();
        uint64_t cap = _static_bytes_cap;
        for (uint32_t i = 0; i < g.count; ++i) {
            page_entry& e = _pages[_order[g.begin + i]];
            const uint64_t need = st[i].code == 0 ? st[i].out_bytes : 0;
            e.bytes_off = cap;
            pd[i].aux_offset = cap;
            pd[i].param0 = static_cast<uint32_t>(std::min<uint64_t>(need, 0xFFFFFFFFull));
            cap += (need + 15) & ~15ull;
        }
        _bytes_cap = cap;
        _d_bytes.ensure(cap + 16, true);      // nothing of this launch() has been written to it yet
        _dev.h2d(d_val_desc + g.begin, pd.data(), g.count * sizeof(p4s_page_desc));
        _dev.sync();      // pd goes out of scope
        return p4s_decode_delta_byte_array(c, d_chunk, in_bytes, d_val_desc + g.begin, g.count, _d_offsets.template as<int64_t>(),
                                           _d_bytes.template as<uint8_t>(), _bytes_cap, _d_scratch.get(), scratch_bytes,
                                           d_val_st + g.begin);
    }
    // bytes reserved for the string groups whose size is known from the encoded sections
    uint64_t _static_bytes_cap = 0;

public:
    // Waits for the launches and turns the per-page statuses into counts and error texts.
    void collect() {
        if (!_launched) launch();
        if (_collected) return;
        const size_t n = _order.size();
        if (n) {
            auto* d_rep_st = _d_status.template as<p4s_page_status>();
            if (_max_rep) _dev.d2h(_rep_st.data(), d_rep_st, n * sizeof(p4s_page_status));
            if (_max_def) _dev.d2h(_def_st.data(), d_rep_st + n, n * sizeof(p4s_page_status));
            _dev.d2h(_val_st.data(), d_rep_st + 2 * n, n * sizeof(p4s_page_status));
        }
        _dev.sync();
        for (size_t li = 0; li < n; ++li) {
            page_entry& e = _pages[_order[li]];
            e.error.clear();
            e.n_rep = e.n_def = e.num_values;         // level_decoder with bit width 0: min(n, num_values) zeros
            // errors in the order the reference meets them: rep reset, def reset, value reset / read
            if (_max_rep) {
                if (_rep_st[li].code) { e.error = page_error_text(_rep_st[li]); continue; }
                e.n_rep = static_cast<uint32_t>(_rep_st[li].values_decoded);
                e.above_rep = static_cast<uint32_t>(_rep_st[li].aux1);
            }
            if (_max_def) {
                if (_def_st[li].code) { e.error = page_error_text(_def_st[li]); continue; }
                e.n_def = static_cast<uint32_t>(_def_st[li].values_decoded);
                e.above_def = static_cast<uint32_t>(_def_st[li].aux1);
            }
            if (_val_st[li].code) { e.error = page_error_text(_val_st[li]); continue; }
            e.n_val = static_cast<uint32_t>(_val_st[li].values_decoded);
            e.bytes_len = _val_st[li].out_bytes;
        }
        _collected = true;
    }

    // ---- positional form of a flat column --------------------------------------------------------------
    // What the reference's record assembly makes of a column without repetition levels, one field at a time
    // (typed_primitive_reader::next, record_reader.hh:449-466: a level below the maximum is a null, any other takes
    // the next dense value), done for the whole chunk by p4s_scatter_by_levels: one value slot per level (slot s of
    // page p at pages()[p].first + s, nulls zero-filled) and a validity bitmap per page.  Fixed-width columns only.
    struct positional_view {
        const void* d_values = nullptr;            // total_slots() elements of slot_elem bytes
        const uint32_t* d_validity = nullptr;      // page p's bitmap starts at word validity_word[p] (bit s = slot s)
        std::vector<uint64_t> validity_word;       // by position in pages(); ~0 for pages without one
        uint64_t rows = 0, nulls = 0;              // levels / nulls of the pages that decoded
    };
    positional_view positional() {
        if constexpr (is_string) throw parquet_exception("positional(): fixed-width columns only");
        if (_max_rep) throw parquet_exception("positional(): the column has repetition levels");
        collect();
        if (_positional_done) return _positional;
        positional_view& v = _positional;
        v.validity_word.assign(_pages.size(), ~0ull);
        std::vector<p4s_page_desc> descs;
        std::vector<size_t> who;
        uint64_t words = 0;
        for (size_t li = 0; li < _order.size(); ++li) {
            page_entry& e = _pages[_order[li]];
            if (!e.error.empty()) continue;
            p4s_page_desc d{};
            d.out_offset = e.first;
            d.num_values = e.n_def;
            d.param0 = _max_def;
            d.param1 = e.n_val;
            d.aux_offset = words;
            v.validity_word[_order[li]] = words;
            words += (static_cast<uint64_t>(e.n_def) + 31) / 32;
            descs.push_back(d);
            who.push_back(_order[li]);
        }
        uint32_t* d_valid = static_cast<uint32_t*>(_d_validity.ensure(std::max<uint64_t>(words, 1) * 4, true));
        v.d_validity = d_valid;
        if (_max_def == 0) {
            // a required column: its dense values are positional already, every slot holds one
            v.d_values = _d_values.get();
            if (words) _dev.check(p4s_memset(_dev.ctx(), d_valid, 0xFF, words * 4), "p4s_memset");
            for (size_t k = 0; k < who.size(); ++k) {
                page_entry& e = _pages[who[k]];
                if (e.n_val < e.n_def) {      // fewer values than the header's levels ask for (column_chunk_reader.hh:134-144)
                    e.error = strcat_all("Invalid or corrupted parquet file: Number of values in batch ", e.n_val,
                                         " is less than indicated by def levels ", e.n_def);
                    v.validity_word[who[k]] = ~0ull;
                    continue;
                }
                v.rows += e.n_def;
            }
            _dev.sync();
        } else if (!descs.empty()) {
            void* d_out = _d_positional.ensure(std::max<uint64_t>(_total_slots, 1) * slot_elem, true);
            v.d_values = d_out;
            device::buffer d_desc(_dev), d_st(_dev);
            auto* dd = static_cast<p4s_page_desc*>(d_desc.ensure(descs.size() * sizeof(p4s_page_desc), true));
            auto* ds = static_cast<p4s_page_status*>(d_st.ensure(descs.size() * sizeof(p4s_page_status), true));
            _dev.h2d(dd, descs.data(), descs.size() * sizeof(p4s_page_desc));
            _dev.check(p4s_scatter_by_levels(_dev.ctx(), _d_def.template as<int32_t>(), _d_values.get(), dd,
                                             static_cast<uint32_t>(descs.size()), slot_elem, d_out, d_valid, ds), "p4s_scatter_by_levels");
            std::vector<p4s_page_status> st(descs.size());
            _dev.d2h(st.data(), ds, st.size() * sizeof(p4s_page_status));
            _dev.sync();
            for (size_t k = 0; k < who.size(); ++k) {
                page_entry& e = _pages[who[k]];
                const uint64_t non_null = st[k].aux0;
                if (st[k].aux1) {             // more non-null levels than values (column_chunk_reader.hh:141-144)
                    e.error = strcat_all("Invalid or corrupted parquet file: Number of values in batch ", e.n_val,
                                         " is less than indicated by def levels ", non_null);
                    v.validity_word[who[k]] = ~0ull;
                    continue;
                }
                v.rows += e.n_def;
                v.nulls += e.n_def - non_null;
            }
        } else {
            v.d_values = _d_values.get();
        }
        _positional_done = true;
        return v;
    }

private:
    positional_view _positional;
public:
    // ---- host copies (column_chunk_reader and tests) ----
    struct host_copy {
        std::vector<int32_t> def, rep;          // total_slots each (empty when the maximum level is 0)
        std::vector<byte> values;               // total_slots * slot_elem
        std::vector<int64_t> offsets;           // total_slots + data pages + 1
        std::vector<byte> str_bytes;
    };
    host_copy download() {
        collect();
        host_copy h;
        const size_t n = _order.size();
        if (n == 0) return h;
        if (_max_def && _total_slots) { h.def.resize(_total_slots); _dev.d2h(h.def.data(), _d_def.get(), _total_slots * 4); }
        if (_max_rep && _total_slots) { h.rep.resize(_total_slots); _dev.d2h(h.rep.data(), _d_rep.get(), _total_slots * 4); }
        if (_d_values.get() && _total_slots) { h.values.resize(_total_slots * slot_elem); _dev.d2h(h.values.data(), _d_values.get(), h.values.size()); }
        if (_d_offsets.get()) { h.offsets.resize(_total_slots + n + 1); _dev.d2h(h.offsets.data(), _d_offsets.get(), h.offsets.size() * 8); }
        if (_d_bytes.get() && _bytes_cap) { h.str_bytes.resize(_bytes_cap); _dev.d2h(h.str_bytes.data(), _d_bytes.get(), _bytes_cap); }
        _dev.sync();
        return h;
    }
};

// ---------------------------------------------------------------------------
// column_chunk_reader<T> (ref: include/parquet4seastar/column_chunk_reader.hh:49-157,
// src/column_chunk_reader.cc:52-163).  The first read_batch decodes the whole chunk on the GPU
// (column_chunk_decoder) and copies the results back; batches are then served with the
// reference's rules: a batch never spans pages, every error is prefixed with the page ordinal.
// A page's load-time and read-time errors both surface at the first batch that touches it.
// ---------------------------------------------------------------------------
template <format::Type::type T>
class column_chunk_reader {
public:
    using output_type = typename value_decoder_traits<T>::output_type;
private:
    using decoder_t = column_chunk_decoder<T>;
    using page_entry = typename decoder_t::page_entry;
    page_reader _source;
    std::unique_ptr<decoder_t> _dec;
    typename decoder_t::host_copy _host;
    bool _initialized = false;      // a data page is loaded
    bool _eof = false;
    int64_t _page_ordinal = -1;
    size_t _next_entry = 0;         // next page of the plan to load
    const page_entry* _page = nullptr;
    uint32_t _def_read = 0, _rep_read = 0, _val_read = 0;
    uint32_t _def_level;
    uint32_t _rep_level;
    std::optional<uint32_t> _type_length;
    format::CompressionCodec::type _codec;
    std::map<int64_t, std::string> _page_errors;
    device::buffer _d_raw;          // the chunk in device memory (P4S_DEVICE_CHUNK)

    void decode_chunk() {
        if (_codec == format::CompressionCodec::SNAPPY && !snappy_host_image_wanted()) {
            // the decompressed pages never leave the device
            _dec = std::make_unique<decoder_t>(snappy_device_chunk(bytes_view(_source.chunk()), device::current(), _def_level, _rep_level),
                                               _def_level, _rep_level, _type_length);
            _codec = format::CompressionCodec::UNCOMPRESSED;
        } else if (_codec == format::CompressionCodec::UNCOMPRESSED && device_chunk_wanted()) {
            device& dev = device::current();
            const bytes& c = _source.chunk();
            _d_raw = device::buffer(dev);
            uint8_t* d = static_cast<uint8_t*>(_d_raw.ensure(c.size() + 16, true));
            dev.h2d(d, c.data(), c.size());
            dev.sync();
            _dec = std::make_unique<decoder_t>(device_chunk_view(d, c.size(), dev, _def_level, _rep_level), _def_level, _rep_level, _type_length);
        } else {
            if (_codec == format::CompressionCodec::SNAPPY) {
                chunk_image img = snappy_chunk_image(bytes_view(_source.chunk()), device::current());
                _source = page_reader(std::move(img.chunk));
                _page_errors = std::move(img.page_errors);
                _codec = format::CompressionCodec::UNCOMPRESSED;
            }
            _dec = std::make_unique<decoder_t>(bytes_view(_source.chunk()), _def_level, _rep_level, _type_length);
            _dec->set_page_errors(_page_errors);
        }
        _dec->launch();
        _host = _dec->download();
    }
    void load_next_page() {                                         // ref: column_chunk_reader.cc:140-163
        ++_page_ordinal;
        if (!_dec) decode_chunk();
        const auto& pages = _dec->pages();
        if (_next_entry >= pages.size()) { _eof = true; return; }
        const page_entry& e = pages[_next_entry++];
        if (!e.error.empty()) throw parquet_exception(e.error);
        if (e.kind == page_entry::data) {
            _page = &e;
            _def_read = _rep_read = _val_read = 0;
            _initialized = true;
        }
    }
    output_type value_at(const page_entry& e, uint32_t i) const {
        using kind = typename decoder_t::value_kind;
        if constexpr (decoder_t::is_string) {
            if (e.vkind == kind::dict) {
                uint32_t idx;
                std::memcpy(&idx, _host.values.data() + (e.first + i) * 4, 4);
                const auto& d = _dec->dictionaries()[e.dict];
                return idx < d.size ? d.host[idx] : output_type();
            }
            if (e.vkind == kind::flba) {
                const uint32_t L = _type_length.value_or(0);
                return output_type(_host.str_bytes.data() + e.bytes_off + static_cast<uint64_t>(i) * L, L);
            }
            const int64_t a = _host.offsets[e.str_slot + i], b = _host.offsets[e.str_slot + i + 1];
            return output_type(_host.str_bytes.data() + e.bytes_off + a, static_cast<size_t>(b - a));
        } else {
            output_type v;
            std::memcpy(&v, _host.values.data() + (e.first + i) * sizeof(output_type), sizeof(output_type));
            return v;
        }
    }
    template <typename LevelT>
    size_t read_batch_internal(size_t n, LevelT def[], LevelT rep[], output_type val[]) {   // ref: column_chunk_reader.hh:101-146
        for (;;) {
            if (_eof) return 0;
            if (!_initialized) { load_next_page(); continue; }
            const page_entry& e = *_page;
            const size_t def_levels_read = std::min<size_t>(n, e.n_def - _def_read);
            const size_t rep_levels_read = std::min<size_t>(n, e.n_rep - _rep_read);
            for (size_t i = 0; i < def_levels_read; ++i)
                def[i] = _def_level ? static_cast<LevelT>(_host.def[e.first + _def_read + i]) : LevelT(0);
            for (size_t i = 0; i < rep_levels_read; ++i)
                rep[i] = _rep_level ? static_cast<LevelT>(_host.rep[e.first + _rep_read + i]) : LevelT(0);
            _def_read += static_cast<uint32_t>(def_levels_read);
            _rep_read += static_cast<uint32_t>(rep_levels_read);
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
            const size_t values_read = std::min<size_t>(values_to_read, e.n_val - _val_read);
            for (size_t i = 0; i < values_read; ++i) val[i] = value_at(e, _val_read + static_cast<uint32_t>(i));
            _val_read += static_cast<uint32_t>(values_read);
            if (values_read != values_to_read)
                throw parquet_exception::corrupted_file(strcat_all("Number of values in batch ", values_read,
                        " is less than indicated by def levels ", values_to_read));
            return def_levels_read;
        }
    }
public:
    explicit column_chunk_reader(page_reader&& source, format::CompressionCodec::type codec, uint32_t def_level,
                                 uint32_t rep_level, std::optional<uint32_t> type_length)
        : _source(std::move(source)), _def_level(def_level), _rep_level(rep_level), _type_length(type_length), _codec(codec) {
        if (codec != format::CompressionCodec::UNCOMPRESSED && codec != format::CompressionCodec::SNAPPY)
            throw parquet_exception("The GPU decode path accepts UNCOMPRESSED and SNAPPY column chunks only");
        if constexpr (T == format::Type::FIXED_LEN_BYTE_ARRAY) {
            if (!_type_length) throw parquet_exception::corrupted_file("type_length not set for FIXED_LEN_BYTE_ARRAY");
        }
    }
    // Read a batch of n (rep, def, value) triplets; the last batch of a page may be smaller than n.
    // Returns the number of triplets read; null values are not written to val.
    template <typename LevelT>
    size_t read_batch(size_t n, LevelT def[], LevelT rep[], output_type val[]) {   // ref: column_chunk_reader.hh:148-157
        try {
            return read_batch_internal(n, def, rep, val);
        } catch (const std::exception& e) {
            _initialized = false;     // the failed page is abandoned; the next call moves on
            throw parquet_exception(strcat_all("Error while reading page number ", _page_ordinal, ": ", e.what()));
        }
    }
    // The device-resident results behind this reader (null before the first read_batch).
    const column_chunk_decoder<T>* decoder() const { return _dec.get(); }
};

// ---------------------------------------------------------------------------
// Record assembly for flat schemas (ref: record_reader::read_one / read_all, include/parquet4seastar/
// record_reader.hh:469-505, over typed_primitive_reader::next, :449-466, for a schema whose fields are all
// primitive: optional or required, no repetition).  The reference pulls one (def, value) pair per field and
// record through futures; here every column chunk of the row group is decoded and brought into positional form
// on the GPU (column_chunk_decoder<T>::positional), and read_all() replays the reference's consumer calls over
// the result: start_record, then per field start_column(name) and append_value(v) or append_null(), end_record.
// A record ends when the first field has no more levels, as in the reference (:496-503); a field whose page fails to
// load throws "In column NAME: ..." before the consumer hears of the column (:440-443), a field that runs out earlier
// throws "No more values buffered" after start_column (and append_null for an optional field).  Fixed-width fields only.
// ---------------------------------------------------------------------------
class flat_record_reader {
    struct field_base {
        std::string name;
        virtual ~field_base() = default;
        virtual void launch() = 0;
        virtual void finish() = 0;
        virtual uint64_t rows() const = 0;
        virtual const void* d_values() const = 0;
        virtual const uint32_t* d_validity() const = 0;
    };
    template <format::Type::type T>
    struct field final : field_base {
        using decoder_t = column_chunk_decoder<T>;
        using output_type = typename decoder_t::output_type;
        std::unique_ptr<decoder_t> dec;
        typename decoder_t::positional_view view;
        std::vector<output_type> values;         // positional, host copy
        std::vector<uint32_t> validity;
        std::vector<std::pair<uint64_t, uint64_t>> spans;      // per decoded page: (first slot, first validity word)
        std::vector<uint32_t> span_rows;
        uint64_t n_rows = 0;
        std::string failure;                     // of the first page that failed (thrown when a record reaches it)
        uint64_t good_rows = 0;                  // records in front of that page
        void launch() override { dec->launch(); }
        void finish() override {
            view = dec->positional();
            const auto& pages = dec->pages();
            for (size_t i = 0; i < pages.size(); ++i) {
                const auto& e = pages[i];
                if (!e.error.empty()) {
                    failure = strcat_all("Error while reading page number ", e.ordinal, ": ", e.error);
                    break;
                }
                if (e.kind != decoder_t::page_entry::data) continue;
                spans.push_back({e.first, view.validity_word[i]});
                span_rows.push_back(e.n_def);
                good_rows += e.n_def;
            }
            n_rows = good_rows;
            device& dev = device::current();
            const uint64_t slots = dec->total_slots();
            values.resize(slots);
            if (slots && view.d_values) dev.d2h(values.data(), view.d_values, slots * sizeof(output_type));
            uint64_t words = 0;
            for (size_t k = 0; k < spans.size(); ++k) words = std::max<uint64_t>(words, spans[k].second + (span_rows[k] + 31) / 32);
            validity.resize(words);
            if (words) dev.d2h(validity.data(), view.d_validity, words * 4);
            dev.sync();
        }
        uint64_t rows() const override { return n_rows; }
        const void* d_values() const override { return view.d_values; }
        const uint32_t* d_validity() const override { return view.d_validity; }
        // record r of the field (records are visited in order: a cursor over the pages)
        size_t cur = 0;
        uint64_t cur_first_row = 0;
        bool optional = false;
        // what current_levels() of the reference does for record r before the consumer hears of the column: a page that
        // fails to load throws here, wrapped with the column's name (record_reader.hh:423-446)
        void check(uint64_t r) const {
            if (r >= good_rows && !failure.empty()) throw parquet_exception(strcat_all("In column ", name, ": ", failure));
        }
        template <typename Consumer>
        void emit(uint64_t r, Consumer& c) {
            if (r >= good_rows) {
                // the field has no more levels: the reference reports level -1, tells the consumer of a null when the field
                // is optional (:477-484) and fails in skip_field / read_field, without the column's name (:251-259, :393-399)
                if (optional) c.append_null();
                throw parquet_exception("No more values buffered");
            }
            while (r >= cur_first_row + span_rows[cur]) { cur_first_row += span_rows[cur]; ++cur; }
            const uint64_t s = r - cur_first_row;
            const bool valid = (validity[spans[cur].second + s / 32] >> (s % 32)) & 1u;
            if (valid) c.append_value(values[spans[cur].first + s]); else c.append_null();
        }
    };
    using any_field = std::variant<field<format::Type::BOOLEAN>*, field<format::Type::INT32>*, field<format::Type::INT64>*,
                                   field<format::Type::INT96>*, field<format::Type::FLOAT>*, field<format::Type::DOUBLE>*>;
    std::vector<std::unique_ptr<field_base>> _fields;
    std::vector<any_field> _typed;
    bool _decoded = false;
public:
    // One field of the schema, in schema order: its column chunk of the row group (UNCOMPRESSED), its maximum
    // definition level (0: required, 1: optional).
    template <format::Type::type T>
    void add_field(std::string name, bytes_view chunk, uint32_t max_def) {
        static_assert(!column_chunk_decoder<T>::is_string, "flat_record_reader: fixed-width fields only");
        auto f = std::make_unique<field<T>>();
        f->name = std::move(name);
        f->optional = max_def > 0;
        f->dec = std::make_unique<column_chunk_decoder<T>>(chunk, max_def, 0, std::nullopt);
        _typed.push_back(f.get());
        _fields.push_back(std::move(f));
    }
    // Decodes every field on the GPU (all launches are enqueued before the first result is waited for).
    void decode() {
        if (_decoded) return;
        for (auto& f : _fields) f->launch();
        for (auto& f : _fields) f->finish();
        _decoded = true;
    }
    size_t fields() const { return _fields.size(); }
    uint64_t records() { decode(); return _fields.empty() ? 0 : _fields[0]->rows(); }
    // Device-resident positional column and validity bitmap of field i (valid after decode()).
    const void* d_values(size_t i) const { return _fields[i]->d_values(); }
    const uint32_t* d_validity(size_t i) const { return _fields[i]->d_validity(); }
    // ref: record_reader::read_all (:494-505)
    template <typename Consumer>
    void read_all(Consumer& c) {
        decode();
        if (_fields.empty()) return;
        // the first field decides when the records end; its own failure surfaces when the records reach it
        const uint64_t n = _fields[0]->rows();
        for (uint64_t r = 0;; ++r) {
            if (r >= n) {
                std::visit([&](auto* f) { if (!f->failure.empty()) throw parquet_exception(strcat_all("In column ", f->name, ": ", f->failure)); }, _typed[0]);
                break;
            }
            c.start_record();
            for (size_t i = 0; i < _fields.size(); ++i) {
                std::visit([&](auto* f) { f->check(r); }, _typed[i]);
                c.start_column(_fields[i]->name);
                std::visit([&](auto* f) { f->emit(r, c); }, _typed[i]);
            }
            c.end_record();
        }
    }
};

}  // namespace parquet4seastar_amd
