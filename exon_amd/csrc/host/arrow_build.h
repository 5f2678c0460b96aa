// arrow_build.h -- producing Arrow C Data Interface arrays from C++ (host memory, owned buffers).
//
// Mirrors what arrow-rs builders + `RecordBatch::try_new_with_options` do for the reference's
// ExonArrayBuilder::try_into_record_batch (exon-common/src/array_builder.rs:20-45): typed column builders
// with validity, and a struct ("record batch") array with an explicit row count so zero-column batches
// still carry their length (`with_row_count(Some(self.len()))`, :29-33).
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/exon_hip.h"

namespace exon {

// ---- owned Arrow arrays / schemas --------------------------------------------------------------------
// A pooled block that the buffers of several arrays live in (host/cram.h: batches built on decoder threads and released on
// the consumer's).  Reference-counted: the struct array AND each of its view children hold a reference, because the Arrow C
// data interface lets a consumer move a child out of its parent and release the parent first -- the block goes back to its
// pool (`put`) when the last holder is released, never while a moved-out child still points into it.
struct SharedBlock {
  std::atomic<int> refs{1};
  void* block = nullptr;
  size_t bytes = 0;
  void (*put)(void*, size_t) = nullptr;
};
inline void block_unref(SharedBlock* b) {
  if (b && b->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
    if (b->block && b->put) b->put(b->block, b->bytes);
    delete b;
  }
}
struct OwnedArray {
  std::vector<void*> bufs;            // malloc'ed, freed on release
  std::vector<const void*> buf_ptrs;  // ArrowArray::buffers
  std::vector<struct ArrowArray*> children;
  struct ArrowArray* dictionary = nullptr;
  SharedBlock* block = nullptr;       // one reference to the block the buffers live in (nullptr: bufs own them)
};
inline void release_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  OwnedArray* o = static_cast<OwnedArray*>(a->private_data);
  for (auto* c : o->children) {
    if (c->release) c->release(c);
    free(c);
  }
  if (o->dictionary) {
    if (o->dictionary->release) o->dictionary->release(o->dictionary);
    free(o->dictionary);
  }
  for (void* b : o->bufs) free(b);
  block_unref(o->block);
  delete o;
  a->release = nullptr;
}
// an array whose buffers live in a shared pooled block: it takes its own reference to the block
inline struct ArrowArray* new_view_array(SharedBlock* block, const void* validity, const void* values, int64_t n, int64_t nulls,
                                         struct ArrowArray* dictionary = nullptr) {
  struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
  OwnedArray* o = new OwnedArray();
  o->buf_ptrs = {validity, values};
  o->dictionary = dictionary;
  if (block) {
    block->refs.fetch_add(1, std::memory_order_relaxed);
    o->block = block;
  }
  memset(a, 0, sizeof *a);
  a->length = n;
  a->null_count = nulls;
  a->n_buffers = 2;
  a->buffers = o->buf_ptrs.data();
  a->dictionary = dictionary;
  a->release = release_array;
  a->private_data = o;
  return a;
}
struct OwnedSchema {
  std::string format, name;
  std::vector<struct ArrowSchema*> children;
  struct ArrowSchema* dictionary = nullptr;
};
inline void release_schema(struct ArrowSchema* s) {
  if (!s || !s->release) return;
  OwnedSchema* o = static_cast<OwnedSchema*>(s->private_data);
  for (auto* c : o->children) {
    if (c->release) c->release(c);
    free(c);
  }
  if (o->dictionary) {
    if (o->dictionary->release) o->dictionary->release(o->dictionary);
    free(o->dictionary);
  }
  delete o;
  s->release = nullptr;
}

inline void make_schema(struct ArrowSchema* s, const char* fmt, const char* name, bool nullable,
                        std::vector<struct ArrowSchema*> kids = {}, struct ArrowSchema* dict = nullptr) {
  OwnedSchema* o = new OwnedSchema{fmt, name, std::move(kids), dict};
  memset(s, 0, sizeof *s);
  s->format = o->format.c_str();
  s->name = o->name.c_str();
  s->flags = nullable ? ARROW_FLAG_NULLABLE : 0;
  s->n_children = (int64_t)o->children.size();
  s->children = o->children.empty() ? nullptr : o->children.data();
  s->dictionary = dict;
  s->release = release_schema;
  s->private_data = o;
}
inline struct ArrowSchema* new_field(const char* fmt, const char* name, bool nullable, struct ArrowSchema* dict = nullptr) {
  struct ArrowSchema* s = static_cast<struct ArrowSchema*>(malloc(sizeof *s));
  make_schema(s, fmt, name, nullable, {}, dict);
  return s;
}

inline void* dup_buf(const void* src, size_t bytes) {
  void* p = malloc(bytes + 64);  // slack: consumers may read whole 16-byte vectors
  if (bytes) memcpy(p, src, bytes);
  memset(static_cast<uint8_t*>(p) + bytes, 0, 64);
  return p;
}

// validity: byte-per-row vector -> Arrow bitmap (nullptr when there are no nulls)
inline void* pack_validity(const std::vector<uint8_t>& valid, int64_t* nulls) {
  *nulls = 0;
  for (uint8_t v : valid) *nulls += !v;
  if (*nulls == 0) return nullptr;
  const size_t n = valid.size();
  uint8_t* bm = static_cast<uint8_t*>(calloc((n + 7) / 8 + 64, 1));
  for (size_t i = 0; i < n; ++i)
    if (valid[i]) bm[i >> 3] |= (uint8_t)(1u << (i & 7));
  return bm;
}

// fixed-width column (values copied); `valid` may be empty (= all valid)
inline void make_primitive(struct ArrowArray* a, const void* values, int64_t n, int elem, const std::vector<uint8_t>& valid,
                           struct ArrowArray* dictionary = nullptr) {
  OwnedArray* o = new OwnedArray();
  int64_t nulls = 0;
  void* vb = valid.empty() ? nullptr : pack_validity(valid, &nulls);
  if (vb) o->bufs.push_back(vb);
  void* data = dup_buf(values, (size_t)n * elem);
  o->bufs.push_back(data);
  o->buf_ptrs = {vb, data};
  o->dictionary = dictionary;
  memset(a, 0, sizeof *a);
  a->length = n;
  a->null_count = nulls;
  a->n_buffers = 2;
  a->buffers = o->buf_ptrs.data();
  a->dictionary = dictionary;
  a->release = release_array;
  a->private_data = o;
}

// Boolean column: values and validity are Arrow bitmaps (byte-per-row inputs)
inline void make_boolean(struct ArrowArray* a, const std::vector<uint8_t>& values, const std::vector<uint8_t>& valid) {
  OwnedArray* o = new OwnedArray();
  const int64_t n = (int64_t)values.size();
  int64_t nulls = 0;
  void* vbits = pack_validity(valid, &nulls);
  uint8_t* bits = static_cast<uint8_t*>(calloc((size_t)(n + 7) / 8 + 64, 1));
  for (int64_t i = 0; i < n; ++i)
    if (values[(size_t)i]) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
  o->bufs = {vbits, bits};
  o->buf_ptrs = {vbits, bits};
  memset(a, 0, sizeof *a);
  a->length = n;
  a->null_count = nulls;
  a->n_buffers = 2;
  a->buffers = o->buf_ptrs.data();
  a->release = release_array;
  a->private_data = o;
}

// Utf8 column
inline void make_utf8(struct ArrowArray* a, const std::vector<int32_t>& offsets, const std::string& data,
                      const std::vector<uint8_t>& valid) {
  OwnedArray* o = new OwnedArray();
  const int64_t n = (int64_t)offsets.size() - 1;
  int64_t nulls = 0;
  void* vb = valid.empty() ? nullptr : pack_validity(valid, &nulls);
  if (vb) o->bufs.push_back(vb);
  void* ob = dup_buf(offsets.data(), offsets.size() * 4);
  void* db = dup_buf(data.data(), data.size());
  o->bufs.push_back(ob);
  o->bufs.push_back(db);
  o->buf_ptrs = {vb, ob, db};
  memset(a, 0, sizeof *a);
  a->length = n < 0 ? 0 : n;
  a->null_count = nulls;
  a->n_buffers = 3;
  a->buffers = o->buf_ptrs.data();
  a->release = release_array;
  a->private_data = o;
}

inline struct ArrowArray* utf8_array(const std::vector<std::string>& strs) {
  std::vector<int32_t> off(1, 0);
  std::string data;
  for (const auto& s : strs) {
    data += s;
    off.push_back((int32_t)data.size());
  }
  struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
  make_utf8(a, off, data, {});
  return a;
}

// A dictionary (Utf8 values) shared by many batches: the values are built once (per slab), every batch's dictionary column gets a
// thin ArrowArray of its own that keeps the values alive (the C data interface gives every exported array its own release).
struct SharedUtf8 {
  std::vector<int32_t> offsets{0};
  std::string data;
  explicit SharedUtf8(const std::vector<std::string>& strs) {
    for (const auto& s : strs) {
      data += s;
      offsets.push_back((int32_t)data.size());
    }
  }
};
struct SharedUtf8Ref {
  std::shared_ptr<const SharedUtf8> values;
  const void* bufs[3];
};
inline struct ArrowArray* shared_utf8_array(const std::shared_ptr<const SharedUtf8>& v) {
  struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
  SharedUtf8Ref* r = new SharedUtf8Ref{v, {nullptr, v->offsets.data(), v->data.data()}};
  memset(a, 0, sizeof *a);
  a->length = (int64_t)v->offsets.size() - 1;
  a->n_buffers = 3;
  a->buffers = r->bufs;
  a->private_data = r;
  a->release = [](struct ArrowArray* x) {
    if (!x || !x->release) return;
    delete static_cast<SharedUtf8Ref*>(x->private_data);
    x->release = nullptr;
  };
  return a;
}

// ---- a batch of VIEW arrays out of ONE allocation ----------------------------------------------------------------------------
// The GPU pipeline cuts a slab into thousands of batches whose columns are views into pinned blocks: with an ArrowArray, a private
// struct and a couple of vectors per column that was ~25 small allocations per batch on the producer's thread, freed on the
// consumer's (the allocator's arenas contend: 12 to 60 ms per 100 M rows from pass to pass).  A BatchArena holds every column's
// ArrowArray, its buffer / child pointers and the references that keep its memory alive; every array releases itself by
// dropping one count, the last one frees the arena.  (Children stay valid when a consumer moves them out: they live in the arena.)
struct BatchArena;
struct ArenaNode {
  struct ArrowArray arr;
  const void* bufs[3];
  struct ArrowArray* kid;
  BatchArena* arena;
};
struct BatchArena {
  std::atomic<int> refs{0};
  int cap = 0, used = 0;
  SharedBlock* blocks[2] = {nullptr, nullptr};                                         // one reference each
  std::shared_ptr<const std::vector<std::shared_ptr<const SharedUtf8>>> dicts;         // the dictionaries' values
  ArenaNode* nodes() { return reinterpret_cast<ArenaNode*>(this + 1); }
};
// n_nodes: every array of the batch, the struct itself included; n_columns: the struct's children (their pointer array lives behind the nodes)
inline BatchArena* new_batch_arena(int n_nodes, int n_columns, SharedBlock* b0, SharedBlock* b1, std::shared_ptr<const std::vector<std::shared_ptr<const SharedUtf8>>> dicts) {
  void* mem = malloc(sizeof(BatchArena) + (size_t)n_nodes * sizeof(ArenaNode) + (size_t)n_columns * sizeof(struct ArrowArray*));
  BatchArena* a = new (mem) BatchArena();
  a->cap = n_nodes;
  a->blocks[0] = b0;
  a->blocks[1] = b1;
  if (b0) b0->refs.fetch_add(1, std::memory_order_relaxed);
  if (b1) b1->refs.fetch_add(1, std::memory_order_relaxed);
  a->dicts = std::move(dicts);
  return a;
}
inline void arena_drop(BatchArena* a) {
  if (a->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  block_unref(a->blocks[0]);
  block_unref(a->blocks[1]);
  a->~BatchArena();
  free(a);
}
inline void release_arena_array(struct ArrowArray* x) {
  if (!x || !x->release) return;
  ArenaNode* nd = static_cast<ArenaNode*>(x->private_data);
  for (int64_t i = 0; i < x->n_children; ++i)
    if (x->children[i] && x->children[i]->release) x->children[i]->release(x->children[i]);
  if (x->dictionary && x->dictionary->release) x->dictionary->release(x->dictionary);
  x->release = nullptr;
  arena_drop(nd->arena);
}
// an array of the arena: up to 3 buffers, at most one child and a dictionary (both arrays of the same arena)
inline struct ArrowArray* arena_array(BatchArena* a, int64_t length, int64_t offset, int64_t nulls, int n_buffers, const void* b0, const void* b1, const void* b2,
                                      struct ArrowArray* kid = nullptr, struct ArrowArray* dict = nullptr) {
  if (a->used >= a->cap) return nullptr;  // (sized by the caller)
  ArenaNode* nd = a->nodes() + a->used++;
  a->refs.fetch_add(1, std::memory_order_relaxed);
  nd->arena = a;
  nd->bufs[0] = b0;
  nd->bufs[1] = b1;
  nd->bufs[2] = b2;
  nd->kid = kid;
  memset(&nd->arr, 0, sizeof nd->arr);
  nd->arr.length = length;
  nd->arr.offset = offset;
  nd->arr.null_count = nulls;
  nd->arr.n_buffers = n_buffers;
  nd->arr.buffers = nd->bufs;
  nd->arr.n_children = kid ? 1 : 0;
  nd->arr.children = kid ? &nd->kid : nullptr;
  nd->arr.dictionary = dict;
  nd->arr.release = release_arena_array;
  nd->arr.private_data = nd;
  return &nd->arr;
}
// the values of dictionary `d` (slab-wide) as an array of the arena
inline struct ArrowArray* arena_dictionary(BatchArena* a, const SharedUtf8& d) {
  return arena_array(a, (int64_t)d.offsets.size() - 1, 0, 0, 3, nullptr, d.offsets.data(), d.data.data());
}
// the batch: a struct array (malloc'ed by the caller, as every batch of the queue) over arrays of the arena
inline void make_struct_of_arena(struct ArrowArray* out, int64_t n, BatchArena* a, const std::vector<struct ArrowArray*>& kids) {
  ArenaNode* nd = a->nodes() + a->used++;  // the parent's own node: its children pointer array lives behind the arena's nodes
  a->refs.fetch_add(1, std::memory_order_relaxed);
  nd->arena = a;
  memset(out, 0, sizeof *out);
  nd->bufs[0] = nullptr;
  struct ArrowArray** kp = reinterpret_cast<struct ArrowArray**>(a->nodes() + a->cap);  // behind the nodes: room for n_columns pointers
  for (size_t i = 0; i < kids.size(); ++i) kp[i] = kids[i];
  nd->kid = nullptr;
  out->length = n;
  out->n_buffers = 1;
  out->buffers = nd->bufs;
  out->n_children = (int64_t)kids.size();
  out->children = kp;
  out->private_data = nd;
  out->release = [](struct ArrowArray* x) {
    if (!x || !x->release) return;
    ArenaNode* p = static_cast<ArenaNode*>(x->private_data);
    for (int64_t i = 0; i < x->n_children; ++i)
      if (x->children[i] && x->children[i]->release) x->children[i]->release(x->children[i]);
    x->release = nullptr;
    arena_drop(p->arena);
  };
}

inline void make_struct(struct ArrowArray* a, int64_t n, std::vector<struct ArrowArray*> kids) {
  OwnedArray* o = new OwnedArray();
  o->children = std::move(kids);
  o->buf_ptrs = {nullptr};
  memset(a, 0, sizeof *a);
  a->length = n;  // explicit row count, as RecordBatchOptions::with_row_count
  a->n_buffers = 1;
  a->buffers = o->buf_ptrs.data();
  a->n_children = (int64_t)o->children.size();
  a->children = o->children.empty() ? nullptr : o->children.data();
  a->release = release_array;
  a->private_data = o;
}

// ---- typed column builders ---------------------------------------------------------------------------
template <typename T>
struct PrimitiveBuilder {
  std::vector<T> values;
  std::vector<uint8_t> valid;  // byte per row
  void append_value(T v) {
    values.push_back(v);
    valid.push_back(1);
  }
  void append_null(T placeholder = T()) {
    values.push_back(placeholder);
    valid.push_back(0);
  }
  size_t len() const { return values.size(); }
  struct ArrowArray* finish(struct ArrowArray* dictionary = nullptr) {
    struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
    make_primitive(a, values.data(), (int64_t)values.size(), (int)sizeof(T), valid, dictionary);
    values.clear();
    valid.clear();
    return a;
  }
};

struct Utf8Builder {
  std::vector<int32_t> offsets{0};
  std::string data;
  std::vector<uint8_t> valid;
  void append_value(const char* p, size_t n) {
    data.append(p, n);
    offsets.push_back((int32_t)data.size());
    valid.push_back(1);
  }
  void append_value(const std::string& s) { append_value(s.data(), s.size()); }
  void append_null() {
    offsets.push_back((int32_t)data.size());
    valid.push_back(0);
  }
  size_t len() const { return offsets.size() - 1; }
  struct ArrowArray* finish() {
    struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
    make_utf8(a, offsets, data, valid);
    offsets.assign(1, 0);
    data.clear();
    valid.clear();
    return a;
  }
};

// append-only string dictionary: ids are stable across batches
struct Dictionary {
  std::vector<std::string> names;
  int32_t lookup_or_insert(const char* p, size_t n) {
    if (names.size() <= 32) {  // contigs, FILTER combos: a handful of short names, a scan beats hashing
      for (size_t i = 0; i < names.size(); ++i)
        if (names[i].size() == n && memcmp(names[i].data(), p, n) == 0) return (int32_t)i;
    } else {  // string INFO values can be many: hash index over everything appended so far
      // `indexed_` = how many of `names` the index covers.  (Comparing index_.size() with names.size() instead rebuilt the
      // whole map on EVERY lookup once `names` held a duplicate -- e.g. a header repeating a ##contig line.)
      if (indexed_ > names.size()) {  // `names` was replaced from outside: start over
        index_.clear();
        indexed_ = 0;
      }
      for (; indexed_ < names.size(); ++indexed_) index_.emplace(names[indexed_], (int32_t)indexed_);  // first occurrence wins
      auto it = index_.find(std::string(p, n));
      if (it != index_.end()) return it->second;
      index_.emplace(std::string(p, n), (int32_t)names.size());
      ++indexed_;
    }
    names.emplace_back(p, n);
    return (int32_t)names.size() - 1;
  }
  std::unordered_map<std::string, int32_t> index_;
  size_t indexed_ = 0;
  int32_t find(const std::string& s) const {
    for (size_t i = 0; i < names.size(); ++i)
      if (names[i] == s) return (int32_t)i;
    return -1;
  }
};

// List<item> column (INFO fields with Number other than 0 / 1: exon-vcf/src/array_builder/info_builder.rs:258-305): Arrow
// layout = validity bitmap + int32 offsets, one child holding the items (themselves nullable: a '.' element is a NULL item)
inline void make_list(struct ArrowArray* a, const std::vector<int32_t>& offsets, const std::vector<uint8_t>& valid,
                      struct ArrowArray* child) {
  OwnedArray* o = new OwnedArray();
  const int64_t n = (int64_t)offsets.size() - 1;
  int64_t nulls = 0;
  void* vb = valid.empty() ? nullptr : pack_validity(valid, &nulls);
  if (vb) o->bufs.push_back(vb);
  void* ob = dup_buf(offsets.data(), offsets.size() * 4);
  o->bufs.push_back(ob);
  o->buf_ptrs = {vb, ob};
  o->children = {child};
  memset(a, 0, sizeof *a);
  a->length = n < 0 ? 0 : n;
  a->null_count = nulls;
  a->n_buffers = 2;
  a->buffers = o->buf_ptrs.data();
  a->n_children = 1;
  a->children = o->children.data();
  a->release = release_array;
  a->private_data = o;
}
template <typename T>
struct ListBuilder {
  std::vector<int32_t> offsets{0};
  std::vector<uint8_t> valid;  // byte per row
  PrimitiveBuilder<T> items;
  void append_null() {
    offsets.push_back((int32_t)items.len());
    valid.push_back(0);
  }
  void close_row() {  // after items.append_value / append_null calls for this row
    offsets.push_back((int32_t)items.len());
    valid.push_back(1);
  }
  size_t len() const { return offsets.size() - 1; }
  struct ArrowArray* finish(struct ArrowArray* item_dictionary = nullptr) {
    struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
    make_list(a, offsets, valid, items.finish(item_dictionary));
    offsets.assign(1, 0);
    valid.clear();
    return a;
  }
};
// List<Utf8> (VCF id / alt: exon-vcf/src/array_builder/lazy_array_builder.rs:169-205)
struct ListUtf8Builder {
  std::vector<int32_t> offsets{0};
  std::vector<uint8_t> valid;
  Utf8Builder items;
  void append_null() {
    offsets.push_back((int32_t)items.len());
    valid.push_back(0);
  }
  void close_row() {
    offsets.push_back((int32_t)items.len());
    valid.push_back(1);
  }
  size_t len() const { return offsets.size() - 1; }
  struct ArrowArray* finish() {
    struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
    make_list(a, offsets, valid, items.finish());
    offsets.assign(1, 0);
    valid.clear();
    return a;
  }
};
// schema field "+l" with its "item" child (item_fmt "f" / "i"; a dictionary-encoded item passes its value type in `dict`)
inline struct ArrowSchema* new_list_field(const char* item_fmt, const char* name, struct ArrowSchema* item_dict = nullptr) {
  struct ArrowSchema* s = static_cast<struct ArrowSchema*>(malloc(sizeof *s));
  make_schema(s, "+l", name, true, {new_field(item_fmt, "item", true, item_dict)});
  return s;
}

// The reference's column-sink contract (exon-common/src/array_builder.rs:20-45)
class ExonArrayBuilder {
 public:
  virtual ~ExonArrayBuilder() = default;
  // Finishes the internal builders and returns the built arrays (projection order).
  virtual std::vector<struct ArrowArray*> finish() = 0;
  virtual size_t len() const = 0;
  bool is_empty() const { return len() == 0; }
  // Struct array with an explicit row count (zero-column batches keep their length).
  void try_into_record_batch(struct ArrowArray* out) {
    const int64_t n = (int64_t)len();
    make_struct(out, n, finish());
  }
};

}  // namespace exon
