// Flat C entry points over the C++ host layer (p4s_host.hh), for callers that cannot
// instantiate C++ templates: the Python parity tests and bench.py drive value_decoder<T>,
// level_decoder, column_chunk_reader<T> and column_chunk_decoder<T> through these.
// Every function returns -1 (or null) on failure and writes the exception text to `err`.
#include <cstdio>
#include <cstring>

#include "p4s_host.hh"

using namespace parquet4seastar_amd;

namespace {

void set_err(char* err, uint64_t err_len, const char* what) {
    if (err && err_len) std::snprintf(err, err_len, "%s", what);
}

template <format::Type::type T>
int64_t value_decode(int encoding, int32_t type_length, const uint8_t* dict_page, uint64_t dict_page_size, int64_t dict_count,
                     const uint8_t* page, uint64_t page_size, const uint64_t* batches, int n_batches,
                     void* out_fixed, uint64_t out_cap, int64_t* out_offsets, uint8_t* out_bytes, uint64_t out_bytes_cap,
                     int64_t* batch_counts) {
    using out_t = typename value_decoder_traits<T>::output_type;
    constexpr bool is_string = std::is_same_v<out_t, bytes>;
    std::optional<uint32_t> tl;
    if (type_length >= 0) tl = static_cast<uint32_t>(type_length);
    value_decoder<T> vd{tl};
    std::vector<out_t> dict;
    if (dict_count >= 0) {      // what load_dictionary_page does (ref: src/column_chunk_reader.cc:127-137)
        dict.resize(static_cast<size_t>(dict_count));
        value_decoder<T> dd{tl};
        dd.reset(bytes_view(dict_page, dict_page_size), format::Encoding::PLAIN);
        size_t got = dd.read_batch(dict.size(), dict.data());
        if (got < dict.size())
            throw parquet_exception::corrupted_file(strcat_all("Unexpected end of dictionary page (expected ", dict.size(), " values, got ", got, ")"));
        vd.reset_dict(dict.data(), dict.size());
    }
    vd.reset(bytes_view(page, page_size), static_cast<format::Encoding::type>(encoding));
    uint64_t total = 0, nbytes = 0;
    if (is_string && out_offsets) out_offsets[0] = 0;
    for (int b = 0; b < n_batches; ++b) {
        std::vector<out_t> tmp(batches[b] ? batches[b] : 1);
        size_t got = vd.read_batch(batches[b], tmp.data());
        if (total + got > out_cap) throw std::runtime_error("value_decode: output capacity exceeded");
        if constexpr (is_string) {
            for (size_t i = 0; i < got; ++i) {
                if (nbytes + tmp[i].size() > out_bytes_cap) throw std::runtime_error("value_decode: byte capacity exceeded");
                std::memcpy(out_bytes + nbytes, tmp[i].data(), tmp[i].size());
                nbytes += tmp[i].size();
                out_offsets[total + i + 1] = static_cast<int64_t>(nbytes);
            }
        } else {
            std::memcpy(static_cast<uint8_t*>(out_fixed) + total * sizeof(out_t), tmp.data(), got * sizeof(out_t));
        }
        total += got;
        if (batch_counts) batch_counts[b] = static_cast<int64_t>(got);
    }
    return static_cast<int64_t>(total);
}

struct chunk_handle {
    virtual ~chunk_handle() = default;
    virtual int64_t read_batch(uint64_t n, int32_t* def, int32_t* rep, void* val_fixed, int64_t* val_offsets, uint8_t* val_bytes,
                               uint64_t val_bytes_cap, uint64_t* values_read) = 0;
    virtual void plan() = 0;
    virtual void launch() = 0;
    virtual void collect(uint64_t* counts) = 0;
    virtual std::string page_error(uint64_t i) = 0;
    virtual const void* device_ptr(int which) = 0;
    virtual int64_t positional(void* values_out, uint8_t* valid_out, uint64_t cap_rows, uint64_t* nulls) = 0;
};

template <format::Type::type T>
struct chunk_impl final : chunk_handle {
    using out_t = typename value_decoder_traits<T>::output_type;
    static constexpr bool is_string = std::is_same_v<out_t, bytes>;
    bytes chunk;
    uint32_t def_level, rep_level;
    std::optional<uint32_t> tl;
    std::unique_ptr<column_chunk_reader<T>> reader;
    std::unique_ptr<column_chunk_decoder<T>> decoder;

    format::CompressionCodec::type codec;
    std::map<int64_t, std::string> page_errors;
    device::buffer d_raw;           // the chunk in device memory (P4S_DEVICE_CHUNK)

    chunk_impl(const uint8_t* p, uint64_t n, uint32_t d, uint32_t r, int32_t type_length, int codec_)
        : chunk(p, n), def_level(d), rep_level(r), codec(static_cast<format::CompressionCodec::type>(codec_)) {
        if (type_length >= 0) tl = static_cast<uint32_t>(type_length);
    }
    int64_t read_batch(uint64_t n, int32_t* def, int32_t* rep, void* val_fixed, int64_t* val_offsets, uint8_t* val_bytes,
                       uint64_t val_bytes_cap, uint64_t* values_read) override {
        if (!reader)
            reader = std::make_unique<column_chunk_reader<T>>(page_reader(bytes_view(chunk)), codec, def_level, rep_level, tl);
        std::vector<out_t> vals(n ? n : 1);
        size_t got = reader->read_batch(n, def, rep, vals.data());
        uint64_t nv = 0;
        for (size_t i = 0; i < got; ++i) nv += static_cast<uint32_t>(def[i]) == def_level;
        if constexpr (is_string) {
            uint64_t nbytes = 0;
            val_offsets[0] = 0;
            for (uint64_t i = 0; i < nv; ++i) {
                if (nbytes + vals[i].size() > val_bytes_cap) throw std::runtime_error("read_batch: byte capacity exceeded");
                std::memcpy(val_bytes + nbytes, vals[i].data(), vals[i].size());
                nbytes += vals[i].size();
                val_offsets[i + 1] = static_cast<int64_t>(nbytes);
            }
        } else {
            std::memcpy(val_fixed, vals.data(), nv * sizeof(out_t));
        }
        *values_read = nv;
        return static_cast<int64_t>(got);
    }
    column_chunk_decoder<T>& dec() {
        if (!decoder) {
            if (codec == format::CompressionCodec::SNAPPY && !snappy_host_image_wanted()) {
                // the pages are decompressed on the device and stay there
                decoder = std::make_unique<column_chunk_decoder<T>>(snappy_device_chunk(bytes_view(chunk), device::current(), def_level, rep_level),
                                                                    def_level, rep_level, tl);
                return *decoder;        // (`chunk` and `codec` stay as they are: a reader made later decompresses for itself)
            }
            if (codec == format::CompressionCodec::UNCOMPRESSED && device_chunk_wanted()) {
                device& dev = device::current();
                d_raw = device::buffer(dev);
                uint8_t* d = static_cast<uint8_t*>(d_raw.ensure(chunk.size() + 16, true));
                dev.h2d(d, chunk.data(), chunk.size());
                dev.sync();
                decoder = std::make_unique<column_chunk_decoder<T>>(device_chunk_view(d, chunk.size(), dev, def_level, rep_level),
                                                                    def_level, rep_level, tl);
                return *decoder;
            }
            if (codec == format::CompressionCodec::SNAPPY) {       // the chunk becomes the uncompressed chunk it stands for
                chunk_image img = snappy_chunk_image(bytes_view(chunk), device::current());
                chunk = std::move(img.chunk);
                page_errors = std::move(img.page_errors);
                codec = format::CompressionCodec::UNCOMPRESSED;
                reader.reset();
            } else if (codec != format::CompressionCodec::UNCOMPRESSED) {
                throw parquet_exception("The GPU decode path accepts UNCOMPRESSED and SNAPPY column chunks only");
            }
            decoder = std::make_unique<column_chunk_decoder<T>>(bytes_view(chunk), def_level, rep_level, tl);
            decoder->set_page_errors(page_errors);
        }
        return *decoder;
    }
    void plan() override { dec().plan(); }
    void launch() override { dec().launch(); }
    void collect(uint64_t* counts) override {
        auto& d = dec();
        d.collect();
        uint64_t levels = 0, values = 0, errors = 0, data_pages = 0, str_bytes = 0;
        for (const auto& e : d.pages()) {
            if (!e.error.empty()) { ++errors; continue; }
            if (e.kind != column_chunk_decoder<T>::page_entry::data) continue;
            ++data_pages;
            levels += std::min(e.n_def, e.n_rep);
            values += e.n_val;
            str_bytes += e.bytes_len;
        }
        counts[0] = d.pages().size(); counts[1] = data_pages; counts[2] = levels; counts[3] = values; counts[4] = errors;
        counts[5] = d.total_slots(); counts[6] = str_bytes;
    }
    std::string page_error(uint64_t i) override { return i < dec().pages().size() ? dec().pages()[i].error : std::string(); }
    // The column in positional form (column_chunk_decoder<T>::positional), copied to the host: the slots of the pages
    // that decoded, in chunk order, one value (nulls 0) and one validity byte per level.  Returns the rows written.
    int64_t positional(void* values_out, uint8_t* valid_out, uint64_t cap_rows, uint64_t* nulls) override {
        if constexpr (is_string) {
            (void)values_out; (void)valid_out; (void)cap_rows; (void)nulls;
            throw parquet_exception("positional(): fixed-width columns only");
        } else {
            auto& d = dec();
            auto v = d.positional();
            device& dev = device::current();
            const uint64_t slots = d.total_slots();
            std::vector<out_t> vals(slots ? slots : 1);
            if (slots && v.d_values) dev.d2h(vals.data(), v.d_values, slots * sizeof(out_t));
            uint64_t words = 0;
            const auto& pages = d.pages();
            for (size_t i = 0; i < pages.size(); ++i)
                if (v.validity_word[i] != ~0ull) words = std::max<uint64_t>(words, v.validity_word[i] + (pages[i].n_def + 31) / 32);
            std::vector<uint32_t> bits(words ? words : 1);
            if (words) dev.d2h(bits.data(), v.d_validity, words * 4);
            dev.sync();
            uint64_t rows = 0;
            for (size_t i = 0; i < pages.size(); ++i) {
                if (v.validity_word[i] == ~0ull) continue;
                const auto& e = pages[i];
                if (rows + e.n_def > cap_rows) throw std::runtime_error("positional: row capacity exceeded");
                std::memcpy(static_cast<out_t*>(values_out) + rows, vals.data() + e.first, e.n_def * sizeof(out_t));
                for (uint32_t k = 0; k < e.n_def; ++k)
                    valid_out[rows + k] = (bits[v.validity_word[i] + k / 32] >> (k % 32)) & 1u;
                rows += e.n_def;
            }
            *nulls = v.nulls;
            return static_cast<int64_t>(rows);
        }
    }
    const void* device_ptr(int which) override {
        auto& d = dec();
        switch (which) {
        case 0: return d.d_def();
        case 1: return d.d_rep();
        case 2: return d.d_values();
        case 3: return d.d_offsets();
        default: return d.d_bytes();
        }
    }
};

}  // namespace

#define P4SH_TYPE_SWITCH(type, CALL)                                                      \
    switch (type) {                                                                       \
    case format::Type::BOOLEAN: CALL(format::Type::BOOLEAN); break;                       \
    case format::Type::INT32: CALL(format::Type::INT32); break;                           \
    case format::Type::INT64: CALL(format::Type::INT64); break;                           \
    case format::Type::INT96: CALL(format::Type::INT96); break;                           \
    case format::Type::FLOAT: CALL(format::Type::FLOAT); break;                           \
    case format::Type::DOUBLE: CALL(format::Type::DOUBLE); break;                         \
    case format::Type::BYTE_ARRAY: CALL(format::Type::BYTE_ARRAY); break;                 \
    case format::Type::FIXED_LEN_BYTE_ARRAY: CALL(format::Type::FIXED_LEN_BYTE_ARRAY); break; \
    default: throw std::invalid_argument("unknown physical type");                        \
    }

extern "C" {

// value_decoder<T>: optional dictionary (a PLAIN dictionary page with dict_count values, dict_count < 0: none),
// reset(page, encoding), then one read_batch per entry of batches[].  Fixed-width types write values to
// out_fixed; BYTE_ARRAY / FIXED_LEN_BYTE_ARRAY write (out_offsets[total + 1], out_bytes).
int64_t p4sh_value_decode(int type, int encoding, int32_t type_length, const uint8_t* dict_page, uint64_t dict_page_size,
                          int64_t dict_count, const uint8_t* page, uint64_t page_size, const uint64_t* batches, int n_batches,
                          void* out_fixed, uint64_t out_cap, int64_t* out_offsets, uint8_t* out_bytes, uint64_t out_bytes_cap,
                          int64_t* batch_counts, char* err, uint64_t err_len) {
    try {
        int64_t r = -1;
#define CALL(T) r = value_decode<T>(encoding, type_length, dict_page, dict_page_size, dict_count, page, page_size, batches, n_batches, \
                                    out_fixed, out_cap, out_offsets, out_bytes, out_bytes_cap, batch_counts)
        P4SH_TYPE_SWITCH(type, CALL)
#undef CALL
        return r;
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return -1;
    }
}

// level_decoder: reset_v1 (v2 == 0; *consumed = its return value) or reset_v2, then one read_batch per batch.
int64_t p4sh_levels_decode(int v2, int encoding, uint32_t max_level, uint32_t num_values, const uint8_t* buf, uint64_t size,
                           const uint64_t* batches, int n_batches, int32_t* out, uint64_t out_cap, int64_t* consumed,
                           char* err, uint64_t err_len) {
    try {
        level_decoder ld(max_level);
        if (v2) { ld.reset_v2(bytes_view(buf, size), num_values); if (consumed) *consumed = static_cast<int64_t>(size); }
        else { size_t c = ld.reset_v1(bytes_view(buf, size), static_cast<format::Encoding::type>(encoding), num_values); if (consumed) *consumed = static_cast<int64_t>(c); }
        uint64_t total = 0;
        for (int b = 0; b < n_batches; ++b) {
            if (total + batches[b] > out_cap) throw std::runtime_error("levels_decode: output capacity exceeded");
            total += ld.read_batch(static_cast<uint32_t>(batches[b]), out + total);
        }
        return static_cast<int64_t>(total);
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return -1;
    }
}

// codec: format::CompressionCodec (0 UNCOMPRESSED, 1 SNAPPY)
void* p4sh_chunk_open_codec(int type, const uint8_t* chunk, uint64_t size, uint32_t def_level, uint32_t rep_level,
                            int32_t type_length, int codec, char* err, uint64_t err_len) {
    try {
        chunk_handle* h = nullptr;
#define CALL(T) h = new chunk_impl<T>(chunk, size, def_level, rep_level, type_length, codec)
        P4SH_TYPE_SWITCH(type, CALL)
#undef CALL
        return h;
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return nullptr;
    }
}

void* p4sh_chunk_open(int type, const uint8_t* chunk, uint64_t size, uint32_t def_level, uint32_t rep_level, int32_t type_length,
                      char* err, uint64_t err_len) {
    return p4sh_chunk_open_codec(type, chunk, size, def_level, rep_level, type_length, 0, err, err_len);
}

void p4sh_chunk_close(void* h) { delete static_cast<chunk_handle*>(h); }

// flat_record_reader for Python: fields are added in schema order (their chunks are copied and kept), read_all
// writes the consumer calls down as text, one record per line: the fields' values (integers in decimal, floating
// point as the 16 hexadecimal digits of its bits widened to double, booleans 0 / 1, INT96 as a/b/c) or "null",
// separated by '|'.  Returns the text's length (the text is cut at cap), or -1 with the error; the records
// delivered before an error are still written.
namespace {
struct flat_handle {
    flat_record_reader reader;
    std::vector<std::unique_ptr<bytes>> chunks;
};
struct text_consumer {
    std::string out;
    bool first = true;
    void start_record() { first = true; }
    void end_record() { out += '\n'; }
    void start_column(const std::string&) { if (!first) out += '|'; first = false; }
    void append_null() { out += "null"; }
    void append_value(const uint8_t& v) { out += v ? '1' : '0'; }
    void append_value(const int32_t& v) { out += std::to_string(v); }
    void append_value(const int64_t& v) { out += std::to_string(v); }
    void append_value(const std::array<int32_t, 3>& v) { out += std::to_string(v[0]) + "/" + std::to_string(v[1]) + "/" + std::to_string(v[2]); }
    void append_bits(double d) { uint64_t u; std::memcpy(&u, &d, 8); char b[20]; std::snprintf(b, sizeof b, "%016llx", static_cast<unsigned long long>(u)); out += b; }
    void append_value(const float& v) { append_bits(static_cast<double>(v)); }
    void append_value(const double& v) { append_bits(v); }
};
}  // namespace

void* p4sh_flat_open(void) { return new (std::nothrow) flat_handle(); }
void p4sh_flat_close(void* h) { delete static_cast<flat_handle*>(h); }
int p4sh_flat_add_field(void* h, int type, const char* name, const uint8_t* chunk, uint64_t size, uint32_t max_def, char* err, uint64_t err_len) {
    try {
        auto* f = static_cast<flat_handle*>(h);
        f->chunks.push_back(std::make_unique<bytes>(chunk, size));
        const bytes_view view(*f->chunks.back());
        switch (type) {
        case format::Type::BOOLEAN: f->reader.add_field<format::Type::BOOLEAN>(name, view, max_def); break;
        case format::Type::INT32: f->reader.add_field<format::Type::INT32>(name, view, max_def); break;
        case format::Type::INT64: f->reader.add_field<format::Type::INT64>(name, view, max_def); break;
        case format::Type::INT96: f->reader.add_field<format::Type::INT96>(name, view, max_def); break;
        case format::Type::FLOAT: f->reader.add_field<format::Type::FLOAT>(name, view, max_def); break;
        case format::Type::DOUBLE: f->reader.add_field<format::Type::DOUBLE>(name, view, max_def); break;
        default: throw parquet_exception("flat_record_reader: fixed-width fields only");
        }
        return 0;
    } catch (const std::exception& e) { set_err(err, err_len, e.what()); return -1; }
}
int64_t p4sh_flat_read_all(void* h, char* out, uint64_t cap, uint64_t* records, char* err, uint64_t err_len) {
    auto* f = static_cast<flat_handle*>(h);
    text_consumer c;
    int64_t rc = 0;
    try {
        f->reader.read_all(c);
        *records = f->reader.records();
    } catch (const std::exception& e) { set_err(err, err_len, e.what()); rc = -1; }
    const uint64_t n = std::min<uint64_t>(c.out.size(), cap);
    std::memcpy(out, c.out.data(), n);
    return rc < 0 ? -1 - static_cast<int64_t>(n) : static_cast<int64_t>(n);
}

// column_chunk_reader<T>::read_batch<int32_t>.  *values_read = number of values written (def == max_def).
int64_t p4sh_chunk_read_batch(void* h, uint64_t n, int32_t* def, int32_t* rep, void* val_fixed, int64_t* val_offsets,
                              uint8_t* val_bytes, uint64_t val_bytes_cap, uint64_t* values_read, char* err, uint64_t err_len) {
    try {
        return static_cast<chunk_handle*>(h)->read_batch(n, def, rep, val_fixed, val_offsets, val_bytes, val_bytes_cap, values_read);
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return -1;
    }
}

// column_chunk_decoder<T>: plan (host walk + upload), launch (kernels only, asynchronous), collect.
// counts[7] = {pages, data pages, levels, values, failed pages, value slots, string bytes}.
int p4sh_chunk_plan(void* h, char* err, uint64_t err_len) {
    try { static_cast<chunk_handle*>(h)->plan(); return 0; } catch (const std::exception& e) { set_err(err, err_len, e.what()); return -1; }
}
int p4sh_chunk_launch(void* h, char* err, uint64_t err_len) {
    try { static_cast<chunk_handle*>(h)->launch(); return 0; } catch (const std::exception& e) { set_err(err, err_len, e.what()); return -1; }
}
int p4sh_chunk_collect(void* h, uint64_t* counts, char* err, uint64_t err_len) {
    try { static_cast<chunk_handle*>(h)->collect(counts); return 0; } catch (const std::exception& e) { set_err(err, err_len, e.what()); return -1; }
}
int p4sh_chunk_page_error(void* h, uint64_t page, char* buf, uint64_t buf_len) {
    try { set_err(buf, buf_len, static_cast<chunk_handle*>(h)->page_error(page).c_str()); return 0; } catch (const std::exception&) { return -1; }
}
int64_t p4sh_chunk_positional(void* h, void* values_out, uint8_t* valid_out, uint64_t cap_rows, uint64_t* nulls, char* err, uint64_t err_len) {
    try { return static_cast<chunk_handle*>(h)->positional(values_out, valid_out, cap_rows, nulls); }
    catch (const std::exception& e) { set_err(err, err_len, e.what()); return -1; }
}
// which: 0 def levels, 1 rep levels, 2 values, 3 offsets, 4 bytes (device pointers, valid after launch)
const void* p4sh_chunk_device_ptr(void* h, int which) {
    try { return static_cast<chunk_handle*>(h)->device_ptr(which); } catch (const std::exception&) { return nullptr; }
}
// The p4s_ctx of the calling thread's device (so that a caller can time or synchronise the launches).
void* p4sh_current_ctx(void) {
    try { return device::current().ctx(); } catch (const std::exception&) { return nullptr; }
}

// The page index of a chunk as page_reader::next_page (the host's own header parse) sees it, in the records
// p4s_walk_page_headers writes (level-section prefixes left at -1).  *count = pages read; returns 0, or -1 with the
// text of the exception that ended the walk (the pages before it are valid).
int p4sh_page_index(const uint8_t* chunk, uint64_t size, p4s_page_info* out, uint64_t max_pages, uint64_t* count,
                    char* err, uint64_t err_len) {
    *count = 0;
    try {
        page_reader src(bytes_view(chunk, size));
        for (;;) {
            const size_t at = src.position();
            std::optional<page> p = src.next_page();
            if (!p) return 0;
            if (*count >= max_pages) { set_err(err, err_len, "page index full"); return -1; }
            const format::PageHeader& h = *p->header;
            p4s_page_info pi{};
            pi.header_offset = at;
            pi.body_offset = src.position() - p->contents.size();
            pi.type = static_cast<int32_t>(h.type);
            pi.uncompressed_page_size = h.uncompressed_page_size;
            pi.compressed_page_size = h.compressed_page_size;
            if (h.isset.data_page_header) {
                pi.flags |= P4S_PAGE_HAS_DATA_HEADER;
                pi.num_values = h.data_page_header.num_values;
                pi.encoding = static_cast<int32_t>(h.data_page_header.encoding);
                pi.definition_level_encoding = static_cast<int32_t>(h.data_page_header.definition_level_encoding);
                pi.repetition_level_encoding = static_cast<int32_t>(h.data_page_header.repetition_level_encoding);
            }
            if (h.isset.dictionary_page_header) {
                pi.flags |= P4S_PAGE_HAS_DICT_HEADER;
                pi.num_values = h.dictionary_page_header.num_values;
                pi.encoding = static_cast<int32_t>(h.dictionary_page_header.encoding);
                if (h.dictionary_page_header.is_sorted) pi.flags |= P4S_PAGE_IS_SORTED;
            }
            if (h.isset.data_page_header_v2) {
                pi.flags |= P4S_PAGE_HAS_DATA_HEADER_V2;
                pi.num_values = h.data_page_header_v2.num_values;
                pi.num_nulls = h.data_page_header_v2.num_nulls;
                pi.num_rows = h.data_page_header_v2.num_rows;
                pi.encoding = static_cast<int32_t>(h.data_page_header_v2.encoding);
                pi.definition_levels_byte_length = h.data_page_header_v2.definition_levels_byte_length;
                pi.repetition_levels_byte_length = h.data_page_header_v2.repetition_levels_byte_length;
                if (!h.data_page_header_v2.isset_is_compressed || h.data_page_header_v2.is_compressed) pi.flags |= P4S_PAGE_IS_COMPRESSED;
            }
            pi.v1_rep_prefix = pi.v1_def_prefix = -1;
            out[(*count)++] = pi;
        }
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return -1;
    }
}

}  // extern "C"
