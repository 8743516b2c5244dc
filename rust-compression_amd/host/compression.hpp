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

template <class T> struct Result {
    bool ok;
    T value;
    CompressionError error;
    static Result Ok(T v) { return Result{true, v, CompressionError::Unexpected}; }
    static Result Err(CompressionError e) { return Result{false, T(), e}; }
};

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
    BZip2Encoder(const BZip2Encoder &) = delete;
    BZip2Encoder &operator=(const BZip2Encoder &) = delete;
    ~BZip2Encoder() { bz_enc_destroy(h_); }

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

// `.collect::<Result<Vec<_>, _>>()`
template <class It> Result<std::vector<uint8_t>> collect(It iter)
{
    std::vector<uint8_t> out;
    for (;;) {
        auto r = iter.next();
        if (!r) break;
        if (!r->ok) return Result<std::vector<uint8_t>>::Err(r->error);
        out.push_back(r->value);
    }
    return Result<std::vector<uint8_t>>::Ok(std::move(out));
}

} // namespace compression
