// Decompresser.hpp — C++ host-side mirror of the reference's operator interface for
// the decompression path, implemented purely on the C ABI (include/zpaqhip.h):
//   Reader / Writer            Reader.cs:7-28, Writer.cs:12-29
//   Decompresser               Decompresser.cs:11-221 (same method names and call order)
//   decompress(Reader*,Writer*) LibZPAQ.cs:65-79
//   error(const char*)         LibZPAQ.cs:22-24: must not return; here it throws zpaq::Error
// Header-only; link with -lzpaqhip.  The whole input is read through Reader::read once,
// all blocks are decoded on the GPU in one call (read-ahead), and the documented call
// sequence is then served from that result.  No CPU decoder is involved.
#pragma once
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "zpaqhip.h"

namespace zpaq {

struct Error : std::runtime_error {
  int code, block, segment;
  Error(const zpaqhip_err &e) : std::runtime_error(e.msg), code(e.code), block(e.block), segment(e.segment) {}
  Error(int c, int b, int s) : std::runtime_error(zpaqhip_strerror(c)), code(c), block(b), segment(s) {}
};

[[noreturn]] inline void error(const zpaqhip_err &e) { throw Error(e); }

class Reader {
 public:
  virtual int get() = 0;                                  // 0..255 or -1 at EOF
  virtual int read(char *buf, int n) {                    // default: n calls of get()
    int i = 0, c;
    while (i < n && (c = get()) >= 0) buf[i++] = (char)c;
    return i;
  }
  virtual ~Reader() {}
};

class Writer {
 public:
  virtual void put(int c) = 0;
  virtual void write(const char *buf, int n) { for (int i = 0; i < n; ++i) put((unsigned char)buf[i]); }
  virtual ~Writer() {}
};

class Decompresser {
 public:
  explicit Decompresser(int device = 0) {
    zpaqhip_err e;
    if (zpaqhip_ctx_create(device, &ctx_, &e)) error(e);
  }
  ~Decompresser() { zpaqhip_ctx_destroy(ctx_); }
  Decompresser(const Decompresser &) = delete;
  Decompresser &operator=(const Decompresser &) = delete;

  void setInput(Reader *in) { in_ = in; loaded_ = false; }                      // Decompresser.cs:22-25

  bool findBlock(double *memptr = nullptr) {                                    // Decompresser.cs:29-58
    load();
    if (b_ + 1 >= (long)blocks_.size()) {
      if (scan_failed_) { scan_failed_ = false; error(scan_err_); }
      return false;
    }
    ++b_;
    s_ = (long)blocks_[b_].first_seg - 1;
    if (memptr) *memptr = blocks_[b_].model_mem;
    return true;
  }
  void hcomp(Writer *out) {                                                     // Decompresser.cs:60-63
    const zpaqhip_block &b = blocks_[b_];
    out->write((const char *)stream_.data() + b.hdr_off, (int)b.hdr_len);
  }
  bool pcomp(Writer *out) {                                                     // Decompresser.cs:155-158
    std::vector<uint8_t> buf(65536 + 2);
    size_t n = 0;
    zpaqhip_err e;
    int rc = zpaqhip_block_pcomp(ctx_, stream_.data(), stream_.size(), (uint32_t)b_, buf.data(), buf.size(), &n, &e);
    if (rc) throw Error(rc, (int)b_, -1);
    if (n) out->write((const char *)buf.data(), (int)n);
    return n != 0;
  }
  bool findFilename(Writer *filename = nullptr) {                               // Decompresser.cs:67-93
    const zpaqhip_block &b = blocks_[b_];
    if (s_ + 1 >= (long)(b.first_seg + b.n_seg)) return false;
    ++s_;
    const zpaqhip_segment &g = segs_[s_];
    if (filename) filename->write((const char *)stream_.data() + g.name_off, (int)g.name_len);
    return true;
  }
  void readComment(Writer *comment = nullptr) {                                 // Decompresser.cs:96-108
    const zpaqhip_segment &g = segs_[s_];
    if (comment) comment->write((const char *)stream_.data() + g.comment_off, (int)g.comment_len);
    pos_ = 0;
  }
  void setOutput(Writer *out) { out_ = out; }                                   // Decompresser.cs:110-113

  bool decompress(int n = -1) {                                                 // Decompresser.cs:121-153
    decode_all();
    const zpaqhip_seg_result &r = res_[s_];
    if (r.status != ZPAQHIP_OK) throw Error(r.status, (int)b_, (int)s_);
    uint64_t left = r.out_len - pos_, take = n < 0 ? left : (left < (uint64_t)n ? left : (uint64_t)n);
    for (uint64_t p = 0; p < take;) {
      int k = (int)(take - p < (1u << 20) ? take - p : (1u << 20));
      if (out_) out_->write((const char *)plain_.data() + r.out_off + pos_ + p, k);
      p += (uint64_t)k;
    }
    pos_ += take;
    return !(n < 0 || take < (uint64_t)n);
  }
  void readSegmentEnd(char *sha1string = nullptr) {                             // Decompresser.cs:163-194
    const zpaqhip_segment &g = segs_[s_];
    if (!sha1string) return;
    sha1string[0] = (char)(g.flags & 1);
    if (g.flags & 1) for (int i = 0; i < 20; ++i) sha1string[i + 1] = (char)g.sha1[i];
  }
  int stat(int) { return 0; }                                                   // Decompresser.cs:196-199
  int buffered() {                                                              // Decompresser.cs:201-204
    if (s_ < 0) return 0;
    return (int)(stream_.size() - (segs_[s_].data_off + segs_[s_].data_len));
  }

 private:
  void load() {
    if (loaded_) return;
    loaded_ = true;
    stream_.clear();
    std::vector<char> buf(1 << 16);
    for (int n; in_ && (n = in_->read(buf.data(), (int)buf.size())) > 0;) stream_.insert(stream_.end(), buf.begin(), buf.begin() + n);
    size_t nb = 0, ns = 0;
    int rc = zpaqhip_scan(stream_.data(), stream_.size(), nullptr, 0, &nb, nullptr, 0, &ns, &scan_err_);
    if (rc) { scan_failed_ = true; nb = ns = 0; }
    blocks_.resize(nb); segs_.resize(ns);
    if (nb && zpaqhip_scan(stream_.data(), stream_.size(), blocks_.data(), nb, &nb, segs_.data(), ns, &ns, &scan_err_)) error(scan_err_);
    b_ = -1; s_ = -1; decoded_ = false;
  }
  void decode_all() {
    if (decoded_) return;
    decoded_ = true;
    // One GPU call for the whole stream (read-ahead); outcomes are kept per segment so an
    // error surfaces only when the caller reaches that segment, as in the reference.
    size_t total = 0, nres = 0;
    zpaqhip_err e;
    res_.assign(segs_.size() ? segs_.size() : 1, zpaqhip_seg_result{});
    int rc = zpaqhip_decompress_segments(ctx_, stream_.data(), stream_.size(), nullptr, 0, &total, res_.data(),
                                         res_.size(), &nres, nullptr, &e);
    if (rc != ZPAQHIP_OK && rc != ZPAQHIP_E_OUTPUT_FULL) error(e);
    plain_.resize(total ? total : 1);
    rc = zpaqhip_decompress_segments(ctx_, stream_.data(), stream_.size(), plain_.data(), total, &total, res_.data(),
                                     res_.size(), &nres, nullptr, &e);
    if (rc) error(e);
  }

  zpaqhip_ctx *ctx_ = nullptr;
  Reader *in_ = nullptr;
  Writer *out_ = nullptr;
  bool loaded_ = false, decoded_ = false, scan_failed_ = false;
  zpaqhip_err scan_err_{};
  std::vector<uint8_t> stream_, plain_;
  std::vector<zpaqhip_block> blocks_;
  std::vector<zpaqhip_segment> segs_;
  std::vector<zpaqhip_seg_result> res_;
  long b_ = -1, s_ = -1;
  uint64_t pos_ = 0;
};

// LibZPAQ.decompress(Reader in, Writer out), LibZPAQ.cs:65-79
inline void decompress(Reader *in, Writer *out, int device = 0) {
  Decompresser d(device);
  d.setInput(in);
  d.setOutput(out);
  while (d.findBlock())
    while (d.findFilename()) {
      d.readComment();
      d.decompress();
      d.readSegmentEnd();
    }
}

}  // namespace zpaq
