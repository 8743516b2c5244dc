// compression.hpp -- C++ mirror of the reference's encode interface for the BZip2 path, over the
// C ABI (include/bz2_mi355x.h).  Header-only; this is what a Rust shim does in Rust (see
// INTEGRATION.md and rust_shim/src/lib.rs).
//
//   reference (Rust)                                     here
//   ---------------------------------------------------  -----------------------------------------
//   enum Action { Run, Flush, Finish }  src/action.rs:8   compression::Action
//   enum CompressionError {..}          src/error.rs:10   compression::CompressionError
//   trait Encoder { fn next(..) }       src/traits/encoder.rs:81-93   compression::Encoder (concept)
//   struct BZip2Encoder                 src/bzip2/encoder.rs:40-159   compression::BZip2Encoder
//   EncodeExt::encode / EncodeIterator  src/traits/encoder.rs:12-79   compression::encode(), EncodeIterator
//   enum BZip2Error {..}                src/bzip2/error.rs:5-11       compression::BZip2Error (+ to_compression_error)
//   struct BZip2Decoder                 src/bzip2/decoder.rs:583-612  compression::BZip2Decoder
//   DecodeExt::decode / DecodeIterator  src/traits/decoder.rs:15-86   compression::decode(), DecodeIterator
//
// Semantics kept: BZip2Encoder(level) throws std::invalid_argument where the reference panics
// (level outside 1..=9); next() returns std::nullopt for None, a Result holding either the byte
// or the error; Default == level 9.
#pragma once
#include "../../include/bz2_mi355x.h"

#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace compression {

enum class Action { Run = BZ_ACTION_RUN, Flush = BZ_ACTION_FLUSH, Finish = BZ_ACTION_FINISH };

enum class CompressionError { DataError, UnexpectedEof, Unexpected };

inline const char *description(CompressionError e)
{ // src/error.rs:34-41
    switch (e) {
    case CompressionError::DataError: return "data integrity error in data";
    case CompressionError::UnexpectedEof: return "file ends unexpectedly";
    default: return "unexpected error";
    }
}

inline CompressionError from_status(int rc)
{
    if (rc == BZ_E_DATA) return CompressionError::DataError;
    if (rc == BZ_E_EOF) return CompressionError::UnexpectedEof;
    return CompressionError::Unexpected; // HIP failures, missing GPU, ... (the encoder's only error, encoder.rs:623)
}

template <class T, class E = CompressionError> struct Result {
    bool ok;
    T value;
    E error;
    static Result Ok(T v) { return Result{true, v, E::Unexpected}; }
    static Result Err(E e) { return Result{false, T(), e}; }
};

// src/bzip2/error.rs:5-11
enum class BZip2Error { DataError, DataErrorMagicFirst, DataErrorMagic, UnexpectedEof, Unexpected };

inline const char *description(BZip2Error e)
{ // src/bzip2/error.rs:31-42
    switch (e) {
    case BZip2Error::DataError: return "data integrity (CRC) error in data";
    case BZip2Error::DataErrorMagicFirst: return "bad magic number (file not created by bzip2)";
    case BZip2Error::DataErrorMagic: return "trailing garbage after EOF ignored";
    case BZip2Error::UnexpectedEof: return "file ends unexpectedly";
    default: return "unexpected error";
    }
}

// impl From<BZip2Error> for CompressionError, src/bzip2/error.rs:45-53
inline CompressionError to_compression_error(BZip2Error e)
{
    if (e == BZip2Error::UnexpectedEof) return CompressionError::UnexpectedEof;
    if (e == BZip2Error::Unexpected) return CompressionError::Unexpected;
    return CompressionError::DataError;
}

inline BZip2Error bzip2_error_from_status(int rc)
{
    switch (rc) {
    case BZ_E_DATA: return BZip2Error::DataError;
    case BZ_E_MAGIC_FIRST: return BZip2Error::DataErrorMagicFirst;
    case BZ_E_MAGIC: return BZip2Error::DataErrorMagic;
    case BZ_E_EOF: return BZip2Error::UnexpectedEof;
    default: return BZip2Error::Unexpected; // HIP failures, missing GPU, ...
    }
}

class BZip2Encoder {
  public:
    using In = uint8_t;
    using Out = uint8_t;
    using Error = CompressionError;

    explicit BZip2Encoder(int level = 9, int device = 0)
    {
        const int rc = bz_enc_create(&h_, level, device);
        if (rc == BZ_E_PARAM) throw std::invalid_argument("invalid level"); // the reference panics
        if (rc != BZ_OK) throw std::runtime_error(bz_strerror(rc));
        buf_.resize(1 << 16);
    }
    // BZip2Encoder::with_devices: the same encoder over several GPUs of this process (bz_enc_create_multi)
    BZip2Encoder(int level, const std::vector<int> &devices)
    {
        const int rc = bz_enc_create_multi(&h_, level, devices.data(), static_cast<int>(devices.size()));
        if (rc == BZ_E_PARAM) throw std::invalid_argument("invalid level or device list");
        if (rc != BZ_OK) throw std::runtime_error(bz_strerror(rc));
        buf_.resize(1 << 16);
    }
    BZip2Encoder(const BZip2Encoder &) = delete;
    BZip2Encoder &operator=(const BZip2Encoder &) = delete;
    ~BZip2Encoder() { bz_enc_destroy(h_); }

    // Self-check (bz_enc_set_verify): every job's blocks are decoded on the device and compared with their input
    // before their bytes are handed out.  stats: blocks checked, jobs redone, redone jobs that failed again, ns.
    BZip2Encoder &verified(bool on = true)
    {
        bz_enc_set_verify(h_, on ? 1 : 0);
        return *this;
    }
    void verify_stats(uint64_t out[4]) { bz_enc_verify_stats(h_, out); }

    // Encoder::next (src/traits/encoder.rs:87-92, src/bzip2/encoder.rs:120-158)
    template <class I, class S> std::optional<Result<uint8_t>> next(I &it, const S &end, Action action)
    {
        if (pos_ == len_) {
            int rc = refill();
            if (rc < 0) return Result<uint8_t>::Err(from_status(rc));
            if (len_ == 0) {
                // pull input in bulk (the reference pulls byte by byte; same bytes, fewer calls)
                for (;;) {
                    chunk_.clear();
                    while (it != end && chunk_.size() < kChunk) {
                        chunk_.push_back(static_cast<uint8_t>(*it));
                        ++it;
                    }
                    if (!chunk_.empty()) {
                        rc = bz_enc_write(h_, chunk_.data(), chunk_.size());
                        if (rc != BZ_OK) return Result<uint8_t>::Err(from_status(rc));
                        if (bz_enc_pending(h_)) break;
                    }
                    if (it == end) {
                        rc = bz_enc_end(h_, static_cast<int>(action));
                        if (rc != BZ_OK) return Result<uint8_t>::Err(from_status(rc));
                        break;
                    }
                }
                rc = refill();
                if (rc < 0) return Result<uint8_t>::Err(from_status(rc));
                if (len_ == 0) return std::nullopt;
            }
        }
        return Result<uint8_t>::Ok(buf_[pos_++]);
    }

  private:
    int refill()
    {
        const long k = bz_enc_read(h_, buf_.data(), buf_.size());
        if (k < 0) return static_cast<int>(k);
        len_ = static_cast<size_t>(k);
        pos_ = 0;
        return 0;
    }
    static constexpr size_t kChunk = 1 << 20;
    bz_enc *h_ = nullptr;
    std::vector<uint8_t> buf_, chunk_;
    size_t pos_ = 0, len_ = 0;
};

// Deflate / zlib / gzip encoders (include/bz2_mi355x.h section 4):
//   Inflater     src/deflate/encoder.rs:92-260   (the reference's name for its Deflate ENCODER)
//   ZlibEncoder  src/zlib/encoder.rs:55-157      GZipEncoder  src/gzip/encoder.rs:50-135
// Action::Run accumulates, Action::Finish produces the stream; Action::Flush writes the bytes so far as a
// byte-aligned segment (Inflater only: the zlib / gzip wrappers refuse it BEFORE pulling any input --
// CompressionError::Unexpected -- because the reference's wrappers end their container at the first None).
template <int Kind> class DeflateFamilyEncoder {
  public:
    using In = uint8_t;
    using Out = uint8_t;
    using Error = CompressionError;

    explicit DeflateFamilyEncoder(int device = 0)
    {
        const int rc = df_enc_create(&h_, Kind, device);
        if (rc != BZ_OK) throw std::runtime_error(bz_strerror(rc));
        buf_.resize(1 << 16);
    }
    // ::with_dict (src/deflate/encoder.rs:134-153, src/zlib/encoder.rs:74-93; GZipEncoder has none)
    static DeflateFamilyEncoder *with_dict(const uint8_t *dict, size_t dict_len, int device = 0)
    {
        auto *e = new DeflateFamilyEncoder(device);
        df_enc_destroy(e->h_);
        e->h_ = nullptr;
        const int rc = df_enc_create_dict(&e->h_, Kind, device, dict, dict_len);
        if (rc != BZ_OK) {
            delete e;
            throw std::runtime_error(bz_strerror(rc));
        }
        return e;
    }
    DeflateFamilyEncoder(const DeflateFamilyEncoder &) = delete;
    DeflateFamilyEncoder &operator=(const DeflateFamilyEncoder &) = delete;
    ~DeflateFamilyEncoder() { df_enc_destroy(h_); }

    template <class I, class S> std::optional<Result<uint8_t>> next(I &it, const S &end, Action action)
    {
        if (pos_ == len_) {
            int rc = refill();
            if (rc < 0) return Result<uint8_t>::Err(from_status(rc));
            if (len_ == 0) {
                // the wrappers do not touch the iterator once their trailer is out (src/zlib/encoder.rs:130-136)
                if (Kind != DF_KIND_DEFLATE && df_enc_finished(h_)) return std::nullopt;
                while (it != end) {
                    chunk_.clear();
                    while (it != end && chunk_.size() < kChunk) {
                        chunk_.push_back(static_cast<uint8_t>(*it));
                        ++it;
                    }
                    rc = df_enc_write(h_, chunk_.data(), chunk_.size());
                    if (rc != BZ_OK) return Result<uint8_t>::Err(from_status(rc));
                }
                rc = df_enc_end(h_, static_cast<int>(action));
                if (rc != BZ_OK) return Result<uint8_t>::Err(from_status(rc));
                rc = refill();
                if (rc < 0) return Result<uint8_t>::Err(from_status(rc));
                if (len_ == 0) return std::nullopt;
            }
        }
        return Result<uint8_t>::Ok(buf_[pos_++]);
    }

  private:
    int refill()
    {
        const long k = df_enc_read(h_, buf_.data(), buf_.size());
        if (k < 0) return static_cast<int>(k);
        len_ = static_cast<size_t>(k);
        pos_ = 0;
        return 0;
    }
    static constexpr size_t kChunk = 1 << 20;
    df_enc *h_ = nullptr;
    std::vector<uint8_t> buf_, chunk_;
    size_t pos_ = 0, len_ = 0;
};
using Inflater = DeflateFamilyEncoder<DF_KIND_DEFLATE>;
using ZlibEncoder = DeflateFamilyEncoder<DF_KIND_ZLIB>;
using GZipEncoder = DeflateFamilyEncoder<DF_KIND_GZIP>;

// EncodeIterator (src/traits/encoder.rs:41-79): a single-pass input range
template <class I, class S, class E> class EncodeIterator {
  public:
    EncodeIterator(I first, S last, E &enc, Action a) : it_(first), end_(last), enc_(enc), action_(a) {}
    std::optional<Result<typename E::Out>> next() { return enc_.next(it_, end_, action_); }

  private:
    I it_;
    S end_;
    E &enc_;
    Action action_;
};

// EncodeExt::encode (src/traits/encoder.rs:25-39)
template <class C, class E> auto encode(const C &container, E &encoder, Action action)
{
    return EncodeIterator<decltype(container.begin()), decltype(container.end()), E>(container.begin(),
                                                                                      container.end(), encoder, action);
}

class BZip2Decoder {
  public:
    using Input = uint8_t;
    using Output = uint8_t;
    using Error = BZip2Error;

    explicit BZip2Decoder(int device = 0) // BZip2Decoder::new, src/bzip2/decoder.rs:588-594
    {
        const int rc = bz_dec_create(&h_, device);
        if (rc != BZ_OK) throw std::runtime_error(bz_strerror(rc));
        buf_.resize(1 << 16);
    }
    BZip2Decoder(const BZip2Decoder &) = delete;
    BZip2Decoder &operator=(const BZip2Decoder &) = delete;
    ~BZip2Decoder() { bz_dec_destroy(h_); }

    // Decoder::next (src/traits/decoder.rs:95-98, src/bzip2/decoder.rs:604-612): the decoded bytes in
    // order, then the Err item if the stream is bad, then None
    template <class I, class S> std::optional<Result<uint8_t, BZip2Error>> next(I &it, const S &end)
    {
        while (pos_ == len_) {
            const long k = bz_dec_read(h_, buf_.data(), buf_.size());
            if (k < 0) {
                if (failed_) return std::nullopt;
                failed_ = true;
                return Result<uint8_t, BZip2Error>::Err(bzip2_error_from_status(static_cast<int>(k)));
            }
            if (k > 0) {
                len_ = static_cast<size_t>(k);
                pos_ = 0;
                break;
            }
            if (ended_) return std::nullopt; // 0 after the end: the clean end
            // nothing ready: hand over more input (the reference pulls bytes on demand; the bytes are the same)
            chunk_.clear();
            while (it != end && chunk_.size() < kChunk) {
                chunk_.push_back(static_cast<uint8_t>(*it));
                ++it;
            }
            if (!chunk_.empty()) {
                const int rc = bz_dec_write(h_, chunk_.data(), chunk_.size());
                if (rc != BZ_OK) return Result<uint8_t, BZip2Error>::Err(bzip2_error_from_status(rc));
            }
            if (it == end) {
                ended_ = true;
                (void)bz_dec_end(h_); // the verdict comes back from bz_dec_read behind the last byte
            }
        }
        return Result<uint8_t, BZip2Error>::Ok(buf_[pos_++]);
    }

  private:
    static constexpr size_t kChunk = 1 << 20;
    bz_dec *h_ = nullptr;
    std::vector<uint8_t> buf_, chunk_;
    size_t pos_ = 0, len_ = 0;
    bool ended_ = false, failed_ = false;
};

// DecodeIterator (src/traits/decoder.rs:45-86)
template <class I, class S, class D> class DecodeIterator {
  public:
    DecodeIterator(I first, S last, D &dec) : it_(first), end_(last), dec_(dec) {}
    std::optional<Result<typename D::Output, typename D::Error>> next() { return dec_.next(it_, end_); }

  private:
    I it_;
    S end_;
    D &dec_;
};

// DecodeExt::decode (src/traits/decoder.rs:27-43)
template <class C, class D> auto decode(const C &container, D &decoder)
{
    return DecodeIterator<decltype(container.begin()), decltype(container.end()), D>(container.begin(),
                                                                                      container.end(), decoder);
}

// `.collect::<Result<Vec<_>, _>>()`
template <class It> auto collect(It iter)
{
    using Item = typename decltype(iter.next())::value_type;
    using Err = decltype(Item::error);
    std::vector<uint8_t> out;
    for (;;) {
        auto r = iter.next();
        if (!r) break;
        if (!r->ok) return Result<std::vector<uint8_t>, Err>::Err(r->error);
        out.push_back(r->value);
    }
    return Result<std::vector<uint8_t>, Err>::Ok(std::move(out));
}

} // namespace compression
