// pymod.cpp — the host binding layer: CPython extension `cramjam_amd._cramjam`.
//
// The reference's host layer is Rust/pyo3 (src/lib.rs, src/io.rs, src/lz4.rs, src/snappy.rs,
// src/exceptions.rs).  Rust is not available in the build image, so this is the same thin layer in
// C++ over the CPython C-API, calling the same extern-"C" boundary (include/cramjam_hip.h) a pyo3
// crate would call (INTEGRATION.md shows that binding).  It keeps, for the hot path only:
//   - cramjam.Buffer            (reference src/io.rs:370-684  RustyBuffer)
//   - BytesType input borrowing (reference src/lib.rs:104-148, src/io.rs:177-299 PythonBuffer)
//   - CompressionError / DecompressionError (reference src/exceptions.rs:6-20)
//   - cramjam.lz4.{compress_block,decompress_block,compress_block_into,decompress_block_into,
//                  compress_block_bound}           (reference src/lz4.rs:78-229)
//   - cramjam.snappy.{compress_raw,decompress_raw,compress_raw_into,decompress_raw_into,
//                  compress_raw_max_len,decompress_raw_len} (reference src/snappy.rs:52-122)
// with the same signatures, defaults, return types and error behaviour.  No codec arithmetic here.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <cerrno>
#include <cstdlib>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <mutex>
#include <new>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/cramjam_hip.h"
#include "xxh32_host.hpp"

namespace {

PyObject* CompressionError = nullptr;
PyObject* DecompressionError = nullptr;

// ---- C++ exceptions never cross into CPython ----------------------------------------------------------------
// Allocation of an attacker-sized result (std::bad_alloc / std::length_error from ByteVec, make_result, the block pool)
// used to unwind through the interpreter's frames and end the process in std::terminate.  Every method the module
// exports now runs under Guarded<>, which turns them into Python exceptions (the reference raises too), and the GIL
// is released through a scope object, so that unwinding out of a GIL-free region re-acquires it first.
struct NoGil {
    PyThreadState* st;
    NoGil() : st(PyEval_SaveThread()) {}
    ~NoGil() { PyEval_RestoreThread(st); }
    NoGil(const NoGil&) = delete;
    NoGil& operator=(const NoGil&) = delete;
};
#undef Py_BEGIN_ALLOW_THREADS
#undef Py_END_ALLOW_THREADS
#define Py_BEGIN_ALLOW_THREADS { NoGil cj_nogil_scope_;
#define Py_END_ALLOW_THREADS }

PyObject* translate_cxx_exception() {           // inside a catch (...) block, GIL held
    try { throw; }
    catch (const std::bad_alloc&) { return PyErr_NoMemory(); }
    catch (const std::length_error&) { PyErr_SetString(PyExc_ValueError, "requested size is too large"); }
    catch (const std::exception& e) { PyErr_SetString(PyExc_RuntimeError, e.what()); }
    catch (...) { PyErr_SetString(PyExc_RuntimeError, "cramjam_amd: unknown C++ exception"); }
    return nullptr;
}
template <auto F> struct Guarded;
template <class... A, PyObject* (*F)(A...)> struct Guarded<F> {
    static PyObject* call(A... a) {
        try { return F(a...); } catch (...) { return translate_cxx_exception(); }
    }
};

// Result buffers are written completely by the device copy, so they are allocated WITHOUT value initialisation (a
// zero-filled 64 MiB std::vector costs ~20 ms of single-threaded page faults + memset — 5x the whole GPU round trip);
// large ones are pre-faulted by a few threads instead.
// Large blocks (>= 4 MiB) are recycled through a small process-wide pool: glibc maps and unmaps such blocks on every
// malloc/free, and faulting 64 MiB of fresh pages in plus unmapping them again costs ~14 ms per call — three times the
// GPU round trip (measured: decompress -> Buffer 18.8 ms vs 4.4 ms into an existing buffer).
struct BigBlockPool {
    static constexpr size_t kMinBytes = 4u << 20, kMaxCachedBytes = 1ull << 30, kMaxCachedBlocks = 8;
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> free_;          // (block, capacity)
    std::unordered_map<void*, size_t> live_;             // capacity of every pooled block handed out
    size_t cached = 0;
    void* get(size_t n) {
        std::lock_guard<std::mutex> g(mu);
        size_t best = free_.size();
        for (size_t i = 0; i < free_.size(); i++)
            if (free_[i].second >= n && free_[i].second <= 2 * n && (best == free_.size() || free_[i].second < free_[best].second)) best = i;
        void* p; size_t cap;
        if (best != free_.size()) { p = free_[best].first; cap = free_[best].second; cached -= cap; free_.erase(free_.begin() + (long)best); }
        else { cap = n; p = std::malloc(n); if (!p) throw std::bad_alloc(); }
        live_[p] = cap;
        return p;
    }
    void put(void* p) {
        std::lock_guard<std::mutex> g(mu);
        auto it = live_.find(p);
        const size_t cap = it->second;
        live_.erase(it);
        if (free_.size() < kMaxCachedBlocks && cached + cap <= kMaxCachedBytes) { free_.emplace_back(p, cap); cached += cap; }
        else std::free(p);
    }
};
inline BigBlockPool& big_pool() { static BigBlockPool* p = new BigBlockPool(); return *p; }   // leaked on purpose: outlives every Buffer

template <class T>
struct DefaultInitAlloc {
    using value_type = T;
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    DefaultInitAlloc() = default;
    template <class U> DefaultInitAlloc(const DefaultInitAlloc<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= BigBlockPool::kMinBytes) return static_cast<T*>(big_pool().get(bytes));
        void* p = std::malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) {
        if (n * sizeof(T) >= BigBlockPool::kMinBytes) big_pool().put(p);
        else std::free(p);
    }
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
    template <class U> bool operator==(const DefaultInitAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const DefaultInitAlloc<U>&) const { return false; }
};
using ByteVec = std::vector<uint8_t, DefaultInitAlloc<uint8_t>>;

void prefault(uint8_t* p, size_t n) {
    if (n < (8u << 20)) return;
    const unsigned t = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    const size_t per = ((n + t - 1) / t + 4095) & ~(size_t)4095;
    for (unsigned k = 0; k < t; k++) {
        const size_t a = k * per, b = std::min(n, a + per);
        if (a >= b) break;
        th.emplace_back([=] { for (size_t i = a; i < b; i += 4096) p[i] = 0; });
    }
    for (auto& x : th) x.join();
}

// n uninitialised bytes; pre-faulted only when the size is one the input can actually produce (`plausible`): a 5-byte
// stream announcing 4 GiB must not make 16 threads touch 4 GiB before the first block has been looked at
ByteVec make_result(size_t n, bool plausible = true) {
    ByteVec v(n);
    if (plausible) prefault(v.data(), n);
    return v;
}
// upper bounds of what n compressed bytes can decode to: an LZ4 length byte stands for at most 255 bytes, a 3-byte
// Snappy copy element for at most 64
inline bool lz4_size_plausible(uint64_t out, uint64_t in_len) { return out <= 255ull * in_len + 64ull; }
inline bool snappy_size_plausible(uint64_t out, uint64_t in_len) { return out <= 22ull * in_len + 64ull; }

// ------------------------------------------------------------------------------------------
// Buffer  (reference src/io.rs:370-684)
// ------------------------------------------------------------------------------------------
struct BufferObject {
    PyObject_HEAD
    ByteVec* vec;                // owned storage (always allocated)
    PyObject* view;              // non-null: zero-copy view of this object (copy=False)
    uint8_t* vptr;               // view pointer / length, re-synchronised on every access
    Py_ssize_t vlen;
    uint64_t pos;                // cursor position (may exceed the length for owned buffers)
};

extern PyTypeObject BufferType;

inline bool Buffer_Check(PyObject* o) { return PyObject_TypeCheck(o, &BufferType); }

// cramjam.File (reference src/io.rs:30-172): a plain read/write file handle usable wherever a BytesType is streamed
struct FileObject {
    PyObject_HEAD
    int fd;
    std::string* path;
};
extern PyTypeObject FileType;
inline bool File_Check(PyObject* o) { return PyObject_TypeCheck(o, &FileType); }
bool file_read_to_end(FileObject* f, std::vector<uint8_t>& out);
bool file_write_all(FileObject* f, const uint8_t* p, size_t n);

// borrowed bytes of any BytesType-like object: Buffer, or anything with the buffer protocol
struct Bytes {
    uint8_t* ptr = nullptr;
    Py_ssize_t len = 0;
    Py_buffer pb{};
    bool have_pb = false;
    BufferObject* buf = nullptr;
    FileObject* file = nullptr;          // set when the object is a File and the caller allowed it
    std::vector<uint8_t> filedata;       // a File INPUT is read (from its position to the end) into here
    ~Bytes() { if (have_pb) PyBuffer_Release(&pb); }
    Bytes() = default;
    Bytes(const Bytes&) = delete;
    Bytes& operator=(const Bytes&) = delete;
};

int buffer_sync_view(BufferObject* self);

uint8_t* buffer_data(BufferObject* b) { return b->view ? b->vptr : b->vec->data(); }
Py_ssize_t buffer_len(BufferObject* b) { return b->view ? b->vlen : (Py_ssize_t)b->vec->size(); }

// reference src/io.rs:273-298 (PythonBuffer::try_from) + src/lib.rs:104-115 (BytesType extraction)
bool get_bytes(PyObject* obj, Bytes& out, int file_mode = 0) {      // file_mode: 0 refuse Files, 1 File input (read now), 2 File output
    if (File_Check(obj)) {
        if (file_mode == 0) {
            PyErr_SetString(PyExc_TypeError, "Converting a File to bytes is not supported, as it'd require reading the entire file "
                                             "into memory; consider using cramjam.Buffer");
            return false;
        }
        out.file = (FileObject*)obj;
        if (file_mode == 1) {
            if (!file_read_to_end(out.file, out.filedata)) return false;
            out.ptr = out.filedata.data(); out.len = (Py_ssize_t)out.filedata.size();
        }
        return true;
    }
    if (Buffer_Check(obj)) {
        BufferObject* b = (BufferObject*)obj;
        if (buffer_sync_view(b) < 0) return false;
        out.buf = b;
        out.ptr = buffer_data(b);
        out.len = buffer_len(b);
        return true;
    }
    if (PyObject_GetBuffer(obj, &out.pb, PyBUF_CONTIG_RO) != 0) {
        PyErr_Clear();
        if (!PyObject_CheckBuffer(obj))
            PyErr_Format(PyExc_TypeError, "argument: failed to extract enum BytesType ('Buffer | File | pybuffer'): "
                                          "'%.100s' object does not support the buffer protocol", Py_TYPE(obj)->tp_name);
        else
            PyErr_SetString(PyExc_BufferError, "Failed to get buffer, is it C contiguous, and shape is not null?");
        return false;
    }
    out.have_pb = true;
    if (out.pb.shape == nullptr) { PyErr_SetString(PyExc_BufferError, "shape is null"); return false; }
    if (!PyBuffer_IsContiguous(&out.pb, 'C')) { PyErr_SetString(PyExc_BufferError, "Buffer is not C contiguous"); return false; }
    out.ptr = (uint8_t*)out.pb.buf;
    out.len = out.pb.len;
    return true;
}

// reference src/io.rs:421-483 ensure_aligned_view
int buffer_sync_view(BufferObject* self) {
    if (!self->view) return 0;
    Bytes b;
    if (!get_bytes(self->view, b)) return -1;
    if (b.ptr != self->vptr || b.len != self->vlen) {
        self->vptr = b.ptr;
        self->vlen = b.len;
        if (self->pos > (uint64_t)b.len) self->pos = (uint64_t)b.len;
    }
    return 0;
}

PyObject* Buffer_new(PyTypeObject* type, PyObject*, PyObject*) {
    BufferObject* self = (BufferObject*)type->tp_alloc(type, 0);
    if (!self) return nullptr;
    self->vec = new ByteVec();
    self->view = nullptr; self->vptr = nullptr; self->vlen = 0; self->pos = 0;
    return (PyObject*)self;
}

PyObject* buffer_from_vec(ByteVec&& v) {   // reference src/io.rs:399-406 From<Vec<u8>>
    BufferObject* b = (BufferObject*)Buffer_new(&BufferType, nullptr, nullptr);
    if (!b) return nullptr;
    *b->vec = std::move(v);
    return (PyObject*)b;
}

void Buffer_dealloc(BufferObject* self) {
    delete self->vec;
    Py_XDECREF(self->view);
    Py_TYPE(self)->tp_free((PyObject*)self);
}

// read everything from a BytesType source at its own cursor (Buffer inputs are consumed from their position)
void read_to_end(Bytes& src, const uint8_t*& p, Py_ssize_t& n) {
    if (src.buf) {
        uint64_t pos = std::min<uint64_t>(src.buf->pos, (uint64_t)src.len);
        p = src.ptr + pos; n = src.len - (Py_ssize_t)pos;
        src.buf->pos = (uint64_t)src.len;
    } else { p = src.ptr; n = src.len; }
}

int Buffer_init(BufferObject* self, PyObject* args, PyObject* kw) {
    static const char* kwl[] = {"data", "copy", nullptr};
    PyObject *data = Py_None, *copy = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|OO", (char**)kwl, &data, &copy)) return -1;
    self->vec->clear(); Py_CLEAR(self->view); self->pos = 0;
    if (data == Py_None) return 0;
    int do_copy = 1;
    if (copy != Py_None) { do_copy = PyObject_IsTrue(copy); if (do_copy < 0) return -1; }
    Bytes b;
    if (!get_bytes(data, b)) return -1;
    if (do_copy) {
        const uint8_t* p; Py_ssize_t n;
        read_to_end(b, p, n);
        self->vec->assign(p, p + n);
    } else {
        Py_INCREF(data);
        self->view = data; self->vptr = b.ptr; self->vlen = b.len;
    }
    return 0;
}

PyObject* Buffer_len(BufferObject* self, PyObject*) {
    if (buffer_sync_view(self) < 0) return nullptr;
    return PyLong_FromSsize_t(buffer_len(self));
}

// Cursor<Vec<u8>>::write semantics: zero-fill a gap, overwrite, extend
void owned_write(BufferObject* self, const uint8_t* p, size_t n) {
    ByteVec& v = *self->vec;
    size_t pos = (size_t)self->pos;
    if (pos > v.size()) v.resize(pos, 0);
    if (pos + n > v.size()) v.resize(pos + n);
    if (n) std::memcpy(v.data() + pos, p, n);
    self->pos = pos + n;
}

PyObject* Buffer_write(BufferObject* self, PyObject* input) {
    if (buffer_sync_view(self) < 0) return nullptr;
    Bytes in;
    if (!get_bytes(input, in, 1)) return nullptr;
    if (self->view) {
        Py_ssize_t room = self->vlen - (Py_ssize_t)std::min<uint64_t>(self->pos, (uint64_t)self->vlen);
        if (in.len > room) { PyErr_SetString(PyExc_OSError, "Too much to write on view"); return nullptr; }
    }
    const uint8_t* p; Py_ssize_t n;
    if ((PyObject*)in.buf == (PyObject*)self) {           // writing a buffer into itself: copy first
        ByteVec tmp(in.ptr + std::min<uint64_t>(self->pos, in.len), in.ptr + in.len);
        if (self->view) { std::memcpy(self->vptr + self->pos, tmp.data(), tmp.size()); self->pos += tmp.size(); }
        else owned_write(self, tmp.data(), tmp.size());
        return PyLong_FromSize_t(tmp.size());
    }
    read_to_end(in, p, n);
    if (self->view) {
        Py_ssize_t room = self->vlen - (Py_ssize_t)self->pos;
        if (n > room) { PyErr_SetString(PyExc_OSError, "failed to write whole buffer"); return nullptr; }
        if (n) std::memcpy(self->vptr + self->pos, p, (size_t)n);
        self->pos += (uint64_t)n;
    } else {
        owned_write(self, p, (size_t)n);
    }
    return PyLong_FromSsize_t(n);
}

PyObject* Buffer_read(BufferObject* self, PyObject* args, PyObject* kw) {
    static const char* kwl[] = {"n_bytes", nullptr};
    PyObject* nobj = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|O", (char**)kwl, &nobj)) return nullptr;
    if (buffer_sync_view(self) < 0) return nullptr;
    Py_ssize_t len = buffer_len(self);
    Py_ssize_t pos = (Py_ssize_t)std::min<uint64_t>(self->pos, (uint64_t)len);
    Py_ssize_t n = len - pos;
    if (nobj != Py_None) {
        Py_ssize_t want = PyLong_AsSsize_t(nobj);
        if (want == -1 && PyErr_Occurred()) return nullptr;
        if (want >= 0) n = std::min(want, n);
    }
    PyObject* r = PyBytes_FromStringAndSize((const char*)buffer_data(self) + pos, n);
    if (r) self->pos = (uint64_t)(pos + n);
    return r;
}

PyObject* Buffer_readinto(BufferObject* self, PyObject* output) {
    if (buffer_sync_view(self) < 0) return nullptr;
    Py_ssize_t len = buffer_len(self);
    Py_ssize_t pos = (Py_ssize_t)std::min<uint64_t>(self->pos, (uint64_t)len);
    Py_ssize_t n = len - pos;
    const uint8_t* src = buffer_data(self) + pos;
    if (Buffer_Check(output)) {
        BufferObject* o = (BufferObject*)output;
        if (o == self) { PyErr_SetString(PyExc_OSError, "cannot readinto self"); return nullptr; }
        if (buffer_sync_view(o) < 0) return nullptr;
        if (o->view) {
            Py_ssize_t room = o->vlen - (Py_ssize_t)std::min<uint64_t>(o->pos, (uint64_t)o->vlen);
            if (n > room) { PyErr_SetString(PyExc_OSError, "failed to write whole buffer"); return nullptr; }
            if (n) std::memcpy(o->vptr + o->pos, src, (size_t)n);
            o->pos += (uint64_t)n;
        } else owned_write(o, src, (size_t)n);
    } else if (File_Check(output)) {
        if (!file_write_all((FileObject*)output, src, (size_t)n)) return nullptr;
    } else {
        Bytes out;
        if (!get_bytes(output, out)) return nullptr;
        if (n > out.len) {     // std::io::copy -> write_all -> WriteZero
            if (out.len) std::memcpy(out.ptr, src, (size_t)out.len);
            self->pos = (uint64_t)(pos + out.len);
            PyErr_SetString(PyExc_OSError, "failed to write whole buffer");
            return nullptr;
        }
        if (n) std::memcpy(out.ptr, src, (size_t)n);
    }
    self->pos = (uint64_t)(pos + n);
    return PyLong_FromSsize_t(n);
}

PyObject* Buffer_seek(BufferObject* self, PyObject* args, PyObject* kw) {
    static const char* kwl[] = {"position", "whence", nullptr};
    Py_ssize_t position; PyObject* wobj = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "n|O", (char**)kwl, &position, &wobj)) return nullptr;
    if (buffer_sync_view(self) < 0) return nullptr;
    long whence = 0;
    if (wobj != Py_None) { whence = PyLong_AsLong(wobj); if (whence == -1 && PyErr_Occurred()) return nullptr; }
    const Py_ssize_t len = buffer_len(self);
    const Py_ssize_t cur = (Py_ssize_t)self->pos;
    Py_ssize_t target;
    switch (whence) {
    case 0:
        target = position;
        if (self->view && (target > len || target < 0))
            return PyErr_Format(PyExc_OSError, "Bad seek: cannot seek outside bounds of unowned buffer, tried to seek from start by %zd which would place it outside of the buffer which has length of %zd.", position, len);
        break;
    case 1:
        target = cur + position;
        if (self->view && (target > len || target < 0))
            return PyErr_Format(PyExc_OSError, "Bad seek: cannot seek outside bounds of unowned buffer, tried to seek from current position %zd by %zd which would place it outside of the buffer which has length of %zd.", cur, target, len);
        break;
    case 2:
        target = len + position;
        if (self->view && (target > len || target < 0))
            return PyErr_Format(PyExc_OSError, "Bad seek: cannot seek outside bounds of unowned buffer, tried to seek from end position by %zd which would place it outside of the buffer which has length of %zd.", position, len);
        break;
    default:
        PyErr_SetString(PyExc_ValueError, "whence should be one of 0: seek from start, 1: seek from current, or 2: seek from end");
        return nullptr;
    }
    if (target < 0) { PyErr_SetString(PyExc_OSError, "invalid seek to a negative or overflowing position"); return nullptr; }
    self->pos = (uint64_t)target;
    return PyLong_FromSsize_t(target);
}

PyObject* Buffer_seekable(BufferObject*, PyObject*) { Py_RETURN_TRUE; }

PyObject* Buffer_tell(BufferObject* self, PyObject*) {
    if (buffer_sync_view(self) < 0) return nullptr;
    return PyLong_FromUnsignedLongLong(self->pos);
}

PyObject* Buffer_set_len(BufferObject* self, PyObject* arg) {
    size_t size = PyLong_AsSize_t(arg);
    if (size == (size_t)-1 && PyErr_Occurred()) return nullptr;
    if (self->view) { PyErr_SetString(PyExc_OSError, "Cannot set length on unowned buffer"); return nullptr; }
    self->vec->resize(size, 0);
    Py_RETURN_NONE;
}

PyObject* Buffer_truncate(BufferObject* self, PyObject*) {
    if (self->view) { PyErr_SetString(PyExc_OSError, "Cannot truncate unowned buffer"); return nullptr; }
    self->vec->clear();
    self->pos = 0;
    Py_RETURN_NONE;
}

PyObject* Buffer_get_view_reference(BufferObject* self, PyObject*) {
    if (!self->view) Py_RETURN_NONE;
    Py_INCREF(self->view);
    return self->view;
}

PyObject* Buffer_get_view_reference_count(BufferObject* self, PyObject*) {
    if (!self->view) Py_RETURN_NONE;
    return PyLong_FromSsize_t(Py_REFCNT(self->view));
}

Py_ssize_t Buffer_sq_length(BufferObject* self) {
    if (buffer_sync_view(self) < 0) return -1;
    return buffer_len(self);
}

int Buffer_contains(BufferObject* self, PyObject* x) {
    Bytes b;
    if (!get_bytes(x, b)) return -1;
    const uint8_t* d = buffer_data(self);
    Py_ssize_t n = buffer_len(self);
    if (b.len == 0) return 0;     // slice::windows(0) panics in the reference; an empty needle is "not found" here
    if (b.len > n) return 0;
    return std::search(d, d + n, b.ptr, b.ptr + b.len) != d + n;
}

PyObject* Buffer_repr(BufferObject* self) {
    if (buffer_sync_view(self) < 0) return nullptr;
    return PyUnicode_FromFormat("cramjam.Buffer<len=%zd>", buffer_len(self));
}

PyObject* Buffer_richcompare(PyObject* a, PyObject* b, int op) {
    if ((op != Py_EQ && op != Py_NE) || !Buffer_Check(a) || !Buffer_Check(b)) Py_RETURN_NOTIMPLEMENTED;
    BufferObject *x = (BufferObject*)a, *y = (BufferObject*)b;
    bool eq = buffer_len(x) == buffer_len(y) && x->pos == y->pos &&
              (buffer_len(x) == 0 || std::memcmp(buffer_data(x), buffer_data(y), (size_t)buffer_len(x)) == 0);
    if ((op == Py_EQ) == eq) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

int Buffer_bool(BufferObject* self) {
    if (buffer_sync_view(self) < 0) return -1;
    return buffer_len(self) > 0;
}

// reference src/io.rs:643-682 __getbuffer__
int Buffer_getbuffer(BufferObject* self, Py_buffer* view, int flags) {
    if (!view) { PyErr_SetString(PyExc_BufferError, "View is null"); return -1; }
    if ((flags & PyBUF_WRITABLE) == PyBUF_WRITABLE) { PyErr_SetString(PyExc_BufferError, "Object is not writable"); view->obj = nullptr; return -1; }
    view->obj = (PyObject*)self;
    Py_INCREF(self);
    view->buf = buffer_data(self);
    view->len = buffer_len(self);
    view->readonly = 0;
    view->itemsize = 1;
    view->format = (flags & PyBUF_FORMAT) == PyBUF_FORMAT ? (char*)"B" : nullptr;
    view->ndim = 1;
    view->shape = (flags & PyBUF_ND) == PyBUF_ND ? &view->len : nullptr;
    view->strides = (flags & PyBUF_STRIDES) == PyBUF_STRIDES ? &view->itemsize : nullptr;
    view->suboffsets = nullptr;
    view->internal = nullptr;
    return 0;
}

PyMethodDef Buffer_methods[] = {
    {"len", (PyCFunction)Guarded<Buffer_len>::call, METH_NOARGS, "Length of the underlying buffer"},
    {"write", (PyCFunction)Guarded<Buffer_write>::call, METH_O, "Write some bytes to the buffer"},
    {"read", (PyCFunction)Guarded<Buffer_read>::call, METH_VARARGS | METH_KEYWORDS, "Read from the buffer at its current position"},
    {"readinto", (PyCFunction)Guarded<Buffer_readinto>::call, METH_O, "Read from the buffer into a bytes-like object"},
    {"seek", (PyCFunction)Guarded<Buffer_seek>::call, METH_VARARGS | METH_KEYWORDS, "Seek; whence 0 start, 1 current, 2 end"},
    {"seekable", (PyCFunction)Guarded<Buffer_seekable>::call, METH_NOARGS, "Always True"},
    {"tell", (PyCFunction)Guarded<Buffer_tell>::call, METH_NOARGS, "Current position"},
    {"set_len", (PyCFunction)Guarded<Buffer_set_len>::call, METH_O, "Set the length; truncates or zero-fills"},
    {"truncate", (PyCFunction)Guarded<Buffer_truncate>::call, METH_NOARGS, "Truncate the buffer"},
    {"get_view_reference", (PyCFunction)Guarded<Buffer_get_view_reference>::call, METH_NOARGS, "Object this Buffer views, or None"},
    {"get_view_reference_count", (PyCFunction)Guarded<Buffer_get_view_reference_count>::call, METH_NOARGS, "Refcount of the viewed object, or None"},
    {nullptr, nullptr, 0, nullptr}};

PySequenceMethods Buffer_as_sequence = {(lenfunc)Buffer_sq_length, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                        (objobjproc)Buffer_contains, nullptr, nullptr};
PyNumberMethods Buffer_as_number = {};
PyBufferProcs Buffer_as_buffer = {(getbufferproc)Buffer_getbuffer, nullptr};

PyTypeObject BufferType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// ------------------------------------------------------------------------------------------
// helpers for the codec functions
// ------------------------------------------------------------------------------------------
PyObject* raise_code(PyObject* exc, int64_t code) {
    if (code == CJ_E_NO_DEVICE || code == CJ_E_OOM || code == CJ_E_BAD_ARG) {
        PyErr_Format(PyExc_RuntimeError, "%s [%s]", cj_strerror(code), cj_last_hip_error());
    } else {
        PyErr_SetString(exc, cj_strerror(code));
    }
    return nullptr;
}

bool opt_size(PyObject* o, bool& has, size_t& v) {
    has = o && o != Py_None;
    if (!has) return true;
    v = PyLong_AsSize_t(o);
    return !(v == (size_t)-1 && PyErr_Occurred());
}

int opt_int(PyObject* o, int dflt) {     // Option<i32> -> -1 for None
    if (!o || o == Py_None) return dflt;
    long v = PyLong_AsLong(o);
    if (v == -1 && PyErr_Occurred()) return -2;
    return (int)v;
}

// ------------------------------------------------------------------------------------------
// cramjam.lz4 block functions (reference src/lz4.rs:78-229)
// ------------------------------------------------------------------------------------------
PyObject* lz4_decompress_block(PyObject*, PyObject* args, PyObject* kw) {       // src/lz4.rs:78-95
    static const char* kwl[] = {"data", "output_len", nullptr};
    PyObject *data, *olen = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O|O", (char**)kwl, &data, &olen)) return nullptr;
    bool has; size_t n = 0;
    if (!opt_size(olen, has, n)) return nullptr;
    Bytes in;
    if (!get_bytes(data, in)) return nullptr;
    ByteVec buf;
    int64_t r;
    if (has) {
        // Some(n): no prefix expected, capacity n, the returned Buffer keeps length n (not truncated, zero tail)
        buf = make_result(n, lz4_size_plausible(n, (uint64_t)in.len));
        Py_BEGIN_ALLOW_THREADS
        r = cj_lz4_block_decompress(in.ptr, (size_t)in.len, buf.data(), n, 0);
        Py_END_ALLOW_THREADS
        if (r < 0) return raise_code(DecompressionError, r);
        if ((size_t)r < n) std::memset(buf.data() + r, 0, n - (size_t)r);
    } else {
        // None: decompress_vec — read the u32-LE prefix, decode, truncate to the decoded length
        int64_t size = cj_lz4_block_prefixed_len(in.ptr, (size_t)in.len);
        if (size < 0) return raise_code(DecompressionError, size);
        if (size > 0x7E000000ll) return raise_code(DecompressionError, size > 0x7FFFFFFFll ? CJ_E_NEG_PREFIX : CJ_E_PREFIX_TOO_BIG);
        // (a prefix no block of this length can fulfil: the decode would fail, say so without allocating gigabytes)
        if (!lz4_size_plausible((uint64_t)size, (uint64_t)in.len)) return raise_code(DecompressionError, CJ_E_CORRUPT);
        buf = make_result((size_t)size);
        Py_BEGIN_ALLOW_THREADS
        r = cj_lz4_block_decompress(in.ptr, (size_t)in.len, buf.data(), (size_t)size, 1);
        Py_END_ALLOW_THREADS
        if (r < 0) return raise_code(DecompressionError, r);
        buf.resize((size_t)r);
    }
    return buffer_from_vec(std::move(buf));
}

bool parse_store_size(PyObject* o, int& prepend) {
    prepend = -1;
    if (o && o != Py_None) { int t = PyObject_IsTrue(o); if (t < 0) return false; prepend = t; }
    return true;
}

PyObject* lz4_compress_block(PyObject*, PyObject* args, PyObject* kw) {         // src/lz4.rs:113-131
    static const char* kwl[] = {"data", "output_len", "mode", "acceleration", "compression", "store_size", nullptr};
    PyObject *data, *olen = Py_None, *mode = Py_None, *accel = Py_None, *comp = Py_None, *store = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O|OOOOO", (char**)kwl, &data, &olen, &mode, &accel, &comp, &store)) return nullptr;
    if (mode != Py_None && !PyUnicode_Check(mode)) { PyErr_SetString(PyExc_TypeError, "argument 'mode': 'str' expected"); return nullptr; }
    int a = opt_int(accel, -1), c = opt_int(comp, -1), prepend;
    if (a == -2 || c == -2 || !parse_store_size(store, prepend)) return nullptr;
    Bytes in;
    if (!get_bytes(data, in)) return nullptr;
    const int pre = prepend != 0;
    size_t bound = cj_lz4_block_compress_bound((size_t)in.len, 0);
    if (bound == 0) return raise_code(CompressionError, CJ_E_INPUT_TOO_LARGE);
    ByteVec buf = make_result(bound + (pre ? 4 : 0));
    int64_t r;
    Py_BEGIN_ALLOW_THREADS
    r = cj_lz4_block_compress(in.ptr, (size_t)in.len, buf.data(), buf.size(), c, a, prepend);
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(CompressionError, r);
    buf.resize((size_t)r);
    return buffer_from_vec(std::move(buf));
}

PyObject* lz4_decompress_block_into(PyObject*, PyObject* args, PyObject* kw) {  // src/lz4.rs:140-173
    static const char* kwl[] = {"input", "output", "output_len", nullptr};
    PyObject *input, *output, *olen = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OO|O", (char**)kwl, &input, &output, &olen)) return nullptr;
    bool has; size_t n = 0;
    if (!opt_size(olen, has, n)) return nullptr;
    Bytes in, out;
    if (!get_bytes(input, in) || !get_bytes(output, out)) return nullptr;
    const int size_stored = !has;
    if (has && (size_t)out.len < n)
        return PyErr_Format(DecompressionError, "output_len set to %zu, but output is less. (%zd)", n, out.len);
    int64_t r;
    Py_BEGIN_ALLOW_THREADS
    r = cj_lz4_block_decompress(in.ptr, (size_t)in.len, out.ptr, (size_t)out.len, size_stored);
    if (r < 0 && r != CJ_E_NO_DEVICE) {
        // fall back to the opposite assumption; the FIRST error wins if that fails too (src/lz4.rs:163-170)
        int64_t r2 = cj_lz4_block_decompress(in.ptr, (size_t)in.len, out.ptr, (size_t)out.len, !size_stored);
        if (r2 >= 0) r = r2;
    }
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(DecompressionError, r);
    return PyLong_FromLongLong(r);
}

PyObject* lz4_compress_block_into(PyObject*, PyObject* args, PyObject* kw) {    // src/lz4.rs:191-216
    static const char* kwl[] = {"data", "output", "mode", "acceleration", "compression", "store_size", nullptr};
    PyObject *data, *output, *mode = Py_None, *accel = Py_None, *comp = Py_None, *store = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OO|OOOO", (char**)kwl, &data, &output, &mode, &accel, &comp, &store)) return nullptr;
    int a = opt_int(accel, -1), c = opt_int(comp, -1), prepend;
    if (a == -2 || c == -2 || !parse_store_size(store, prepend)) return nullptr;
    Bytes in, out;
    if (!get_bytes(data, in) || !get_bytes(output, out)) return nullptr;
    int64_t r;
    Py_BEGIN_ALLOW_THREADS
    r = cj_lz4_block_compress(in.ptr, (size_t)in.len, out.ptr, (size_t)out.len, c, a, prepend);
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(CompressionError, r);
    return PyLong_FromLongLong(r);
}

PyObject* lz4_compress_block_bound(PyObject*, PyObject* src) {                  // src/lz4.rs:226-229
    Bytes in;
    if (!get_bytes(src, in)) return nullptr;
    return PyLong_FromSize_t(cj_lz4_block_compress_bound((size_t)in.len, 1));
}

// ------------------------------------------------------------------------------------------
// cramjam.snappy raw functions (reference src/snappy.rs:52-122)
// ------------------------------------------------------------------------------------------
PyObject* snappy_decompress_raw(PyObject*, PyObject* args, PyObject* kw) {      // src/snappy.rs:52-60
    static const char* kwl[] = {"data", "output_len", nullptr};
    PyObject *data, *olen = Py_None;    // output_len is accepted and ignored, as in the reference
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O|O", (char**)kwl, &data, &olen)) return nullptr;
    Bytes in;
    if (!get_bytes(data, in)) return nullptr;
    if (in.len == 0) return raise_code(DecompressionError, CJ_E_SNAPPY_EMPTY);
    int64_t n = cj_snappy_raw_decompress_len(in.ptr, (size_t)in.len);
    if (n < 0) return raise_code(DecompressionError, n);
    if (!snappy_size_plausible((uint64_t)n, (uint64_t)in.len)) return raise_code(DecompressionError, CJ_E_SNAPPY_CORRUPT);
    ByteVec buf = make_result((size_t)n);
    int64_t r;
    Py_BEGIN_ALLOW_THREADS
    r = cj_snappy_raw_decompress(in.ptr, (size_t)in.len, buf.data(), buf.size());
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(DecompressionError, r);
    buf.resize((size_t)r);
    return buffer_from_vec(std::move(buf));
}

PyObject* snappy_compress_raw(PyObject*, PyObject* args, PyObject* kw) {        // src/snappy.rs:70-78
    static const char* kwl[] = {"data", "output_len", nullptr};
    PyObject *data, *olen = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O|O", (char**)kwl, &data, &olen)) return nullptr;
    Bytes in;
    if (!get_bytes(data, in)) return nullptr;
    size_t cap = cj_snappy_raw_max_compress_len((size_t)in.len);
    if (cap == 0) return raise_code(CompressionError, CJ_E_SNAPPY_TOO_BIG);
    ByteVec buf = make_result(cap);
    int64_t r;
    Py_BEGIN_ALLOW_THREADS
    r = cj_snappy_raw_compress(in.ptr, (size_t)in.len, buf.data(), cap);
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(CompressionError, r);
    buf.resize((size_t)r);
    return buffer_from_vec(std::move(buf));
}

PyObject* snappy_xxx_raw_into(PyObject* args, PyObject* kw, bool compress) {    // src/snappy.rs:93-108
    static const char* kwl[] = {"input", "output", nullptr};
    PyObject *input, *output;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OO", (char**)kwl, &input, &output)) return nullptr;
    Bytes in, out;
    if (!get_bytes(input, in) || !get_bytes(output, out)) return nullptr;
    int64_t r;
    Py_BEGIN_ALLOW_THREADS
    r = compress ? cj_snappy_raw_compress(in.ptr, (size_t)in.len, out.ptr, (size_t)out.len)
                 : cj_snappy_raw_decompress(in.ptr, (size_t)in.len, out.ptr, (size_t)out.len);
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(compress ? CompressionError : DecompressionError, r);
    return PyLong_FromLongLong(r);
}
PyObject* snappy_compress_raw_into(PyObject*, PyObject* a, PyObject* k) { return snappy_xxx_raw_into(a, k, true); }
PyObject* snappy_decompress_raw_into(PyObject*, PyObject* a, PyObject* k) { return snappy_xxx_raw_into(a, k, false); }

PyObject* snappy_compress_raw_max_len(PyObject*, PyObject* data) {              // src/snappy.rs:112-115
    Bytes in;
    if (!get_bytes(data, in)) return nullptr;
    return PyLong_FromSize_t(cj_snappy_raw_max_compress_len((size_t)in.len));
}

PyObject* snappy_decompress_raw_len(PyObject*, PyObject* data) {                // src/snappy.rs:119-122
    Bytes in;
    if (!get_bytes(data, in)) return nullptr;
    int64_t n = cj_snappy_raw_decompress_len(in.ptr, (size_t)in.len);
    if (n < 0) return raise_code(DecompressionError, n);
    return PyLong_FromLongLong(n);
}

// ------------------------------------------------------------------------------------------
// cramjam.snappy framed functions (reference src/snappy.rs:22-42,80-91 via the generic! macro, src/lib.rs:211-296)
// ------------------------------------------------------------------------------------------
// One framed codec: size query (header arithmetic only) + the device call.  `level` is only used by LZ4 compression.
struct Framed {
    int64_t (*need)(const Bytes& in, bool compress);
    int64_t (*run)(const Bytes& in, uint8_t* out, size_t cap, bool compress, int level);
};

int64_t snappy_need(const Bytes& in, bool compress) {
    if (compress) return (int64_t)cj_snappy_frame_max_compress_len((size_t)in.len);
    int64_t d = cj_snappy_frame_decompress_len(in.ptr, (size_t)in.len);
    // header-level error: let the decoder name the FIRST error in stream order (an earlier piece may be corrupt too)
    if (d < 0) { int64_t r = cj_snappy_frame_decompress(in.ptr, (size_t)in.len, nullptr, 0); return r < 0 ? r : d; }
    return d;
}
int64_t snappy_run(const Bytes& in, uint8_t* out, size_t cap, bool compress, int) {
    return compress ? cj_snappy_frame_compress(in.ptr, (size_t)in.len, out, cap) : cj_snappy_frame_decompress(in.ptr, (size_t)in.len, out, cap);
}
int64_t lz4f_need(const Bytes& in, bool compress) {
    return compress ? (int64_t)cj_lz4_frame_compress_bound((size_t)in.len) : cj_lz4_frame_decompress_bound(in.ptr, (size_t)in.len);
}
int64_t lz4f_run(const Bytes& in, uint8_t* out, size_t cap, bool compress, int level) {
    return compress ? cj_lz4_frame_compress(in.ptr, (size_t)in.len, out, cap, level) : cj_lz4_frame_decompress(in.ptr, (size_t)in.len, out, cap);
}
const Framed kSnappyFramed = { snappy_need, snappy_run };
const Framed kLz4Framed = { lz4f_need, lz4f_run };

// (data, [level,] output_len=None) -> Buffer         generic! macro, src/lib.rs:213-235
PyObject* framed_call(const Framed& fc, PyObject* args, PyObject* kw, bool compress, bool with_level) {
    static const char* kwl2[] = {"data", "output_len", nullptr};
    static const char* kwl3[] = {"data", "level", "output_len", nullptr};
    PyObject *data, *olen = Py_None, *lvl = Py_None;
    if (with_level) { if (!PyArg_ParseTupleAndKeywords(args, kw, "O|OO", (char**)kwl3, &data, &lvl, &olen)) return nullptr; }
    else if (!PyArg_ParseTupleAndKeywords(args, kw, "O|O", (char**)kwl2, &data, &olen)) return nullptr;
    bool has; size_t n = 0;
    if (!opt_size(olen, has, n)) return nullptr;
    const int level = opt_int(lvl, -1);
    if (level == -2) return nullptr;
    Bytes in;
    if (!get_bytes(data, in, 1)) return nullptr;
    PyObject* exc = compress ? CompressionError : DecompressionError;
    int64_t need, r;
    ByteVec buf;
    Py_BEGIN_ALLOW_THREADS
    need = fc.need(in, compress);
    if (need >= 0) {
        // generic!: vec![0; output_len] under a Cursor at 0 -> the result is never shorter than output_len (zero tail)
        buf = make_result(std::max((size_t)need, has ? n : (size_t)0));
        r = fc.run(in, buf.data(), buf.size(), compress, level);
        if (r >= 0 && has && (size_t)r < n) std::memset(buf.data() + r, 0, n - (size_t)r);
    } else r = need;
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(exc, r);
    buf.resize(std::max((size_t)r, has ? n : (size_t)0));
    return buffer_from_vec(std::move(buf));
}

// (input, output[, level]) -> int                    generic! macro, src/lib.rs:237-296
PyObject* framed_into(const Framed& fc, PyObject* args, PyObject* kw, bool compress, bool with_level) {
    static const char* kwl2[] = {"input", "output", nullptr};
    static const char* kwl3[] = {"input", "output", "level", nullptr};
    PyObject *input, *output, *lvl = Py_None;
    if (with_level) { if (!PyArg_ParseTupleAndKeywords(args, kw, "OO|O", (char**)kwl3, &input, &output, &lvl)) return nullptr; }
    else if (!PyArg_ParseTupleAndKeywords(args, kw, "OO", (char**)kwl2, &input, &output)) return nullptr;
    const int level = opt_int(lvl, -1);
    if (level == -2) return nullptr;
    Bytes in, out;
    if (!get_bytes(input, in, 1) || !get_bytes(output, out, 2)) return nullptr;
    PyObject* exc = compress ? CompressionError : DecompressionError;
    int64_t r;
    if (out.buf || out.file) {
        // Buffer output: a Cursor<Vec<u8>> written at its position, growing as needed (views cannot grow); File output:
        // written at the file's position
        ByteVec tmp;
        Py_BEGIN_ALLOW_THREADS
        r = fc.need(in, compress);
        if (r >= 0) {
            tmp = make_result((size_t)r);
            r = fc.run(in, tmp.data(), tmp.size(), compress, level);
        }
        Py_END_ALLOW_THREADS
        if (r < 0) return raise_code(exc, r);
        if (out.file) {
            if (!file_write_all(out.file, tmp.data(), (size_t)r)) return nullptr;
            return PyLong_FromLongLong(r);
        }
        BufferObject* b = out.buf;
        if (b->view) {
            const uint64_t pos = std::min<uint64_t>(b->pos, (uint64_t)b->vlen);
            if ((uint64_t)r > (uint64_t)b->vlen - pos) return raise_code(exc, CJ_E_FRAME_WRITE);
            if (r) std::memcpy(b->vptr + pos, tmp.data(), (size_t)r);
            b->pos = pos + (uint64_t)r;
        } else {
            owned_write(b, tmp.data(), (size_t)r);
        }
        return PyLong_FromLongLong(r);
    }
    Py_BEGIN_ALLOW_THREADS
    r = fc.run(in, out.ptr, (size_t)out.len, compress, level);
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(exc, r);
    return PyLong_FromLongLong(r);
}

// cramjam.snappy framed functions (reference src/snappy.rs:22-42,80-91)
PyObject* snappy_compress(PyObject*, PyObject* a, PyObject* k) { return framed_call(kSnappyFramed, a, k, true, false); }
PyObject* snappy_decompress(PyObject*, PyObject* a, PyObject* k) { return framed_call(kSnappyFramed, a, k, false, false); }
PyObject* snappy_compress_into(PyObject*, PyObject* a, PyObject* k) { return framed_into(kSnappyFramed, a, k, true, false); }
PyObject* snappy_decompress_into(PyObject*, PyObject* a, PyObject* k) { return framed_into(kSnappyFramed, a, k, false, false); }
// cramjam.lz4 frame functions (reference src/lz4.rs:18-66)
PyObject* lz4_compress(PyObject*, PyObject* a, PyObject* k) { return framed_call(kLz4Framed, a, k, true, true); }
PyObject* lz4_decompress(PyObject*, PyObject* a, PyObject* k) { return framed_call(kLz4Framed, a, k, false, false); }
PyObject* lz4_compress_into(PyObject*, PyObject* a, PyObject* k) { return framed_into(kLz4Framed, a, k, true, true); }
PyObject* lz4_decompress_into(PyObject*, PyObject* a, PyObject* k) { return framed_into(kLz4Framed, a, k, false, false); }

// ------------------------------------------------------------------------------------------
// cramjam.File (reference src/io.rs:30-172)
// ------------------------------------------------------------------------------------------
PyObject* os_error(const char* what) { return PyErr_Format(PyExc_OSError, "%s: %s", what, std::strerror(errno)); }

bool file_read_to_end(FileObject* f, std::vector<uint8_t>& out) {
    out.clear();
    uint8_t chunk[1 << 16];
    for (;;) {
        const ssize_t r = ::read(f->fd, chunk, sizeof chunk);
        if (r < 0) { if (errno == EINTR) continue; os_error("read"); return false; }
        if (r == 0) return true;
        out.insert(out.end(), chunk, chunk + r);
    }
}
bool file_write_all(FileObject* f, const uint8_t* p, size_t n) {
    while (n) {
        const ssize_t r = ::write(f->fd, p, n);
        if (r < 0) { if (errno == EINTR) continue; os_error("write"); return false; }
        p += r; n -= (size_t)r;
    }
    return true;
}

PyObject* File_new(PyTypeObject* type, PyObject*, PyObject*) {
    FileObject* self = (FileObject*)type->tp_alloc(type, 0);
    if (!self) return nullptr;
    self->fd = -1; self->path = new std::string();
    return (PyObject*)self;
}
int File_init(FileObject* self, PyObject* args, PyObject* kw) {
    static const char* kwl[] = {"path", "read", "write", "truncate", "append", nullptr};
    const char* path; PyObject *rd = Py_None, *wr = Py_None, *tr = Py_None, *ap = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "s|OOOO", (char**)kwl, &path, &rd, &wr, &tr, &ap)) return -1;
    auto flag = [](PyObject* o, bool dflt, bool& ok) { if (o == Py_None) return dflt; int t = PyObject_IsTrue(o); if (t < 0) ok = false; return t > 0; };
    bool ok = true;
    const bool r = flag(rd, true, ok), w = flag(wr, true, ok), t = flag(tr, false, ok), a = flag(ap, false, ok);
    if (!ok) return -1;
    int fl = (r && (w || a)) ? O_RDWR : (w || a) ? O_WRONLY : O_RDONLY;
    fl |= O_CREAT | O_CLOEXEC;            // create if it doesn't exist, but open if it does
    if (t) fl |= O_TRUNC;
    if (a) fl |= O_APPEND;
    if (self->fd >= 0) ::close(self->fd);
    self->fd = ::open(path, fl, 0666);
    if (self->fd < 0) { PyErr_SetFromErrnoWithFilename(PyExc_OSError, path); return -1; }
    *self->path = path;
    return 0;
}
void File_dealloc(FileObject* self) {
    if (self->fd >= 0) ::close(self->fd);
    delete self->path;
    Py_TYPE(self)->tp_free((PyObject*)self);
}
PyObject* File_write(FileObject* self, PyObject* input) {
    Bytes in;
    if ((PyObject*)self == input) { PyErr_SetString(PyExc_OSError, "cannot write a file into itself"); return nullptr; }
    if (!get_bytes(input, in, 1)) return nullptr;
    const uint8_t* p; Py_ssize_t n;
    read_to_end(in, p, n);                // Buffer inputs are consumed from their position
    if (!file_write_all(self, p, (size_t)n)) return nullptr;
    return PyLong_FromSsize_t(n);
}
PyObject* File_read(FileObject* self, PyObject* args, PyObject* kw) {
    static const char* kwl[] = {"n_bytes", nullptr};
    PyObject* nobj = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|O", (char**)kwl, &nobj)) return nullptr;
    if (nobj == Py_None) {
        std::vector<uint8_t> all;
        if (!file_read_to_end(self, all)) return nullptr;
        return PyBytes_FromStringAndSize((const char*)all.data(), (Py_ssize_t)all.size());
    }
    const Py_ssize_t want = PyLong_AsSsize_t(nobj);
    if (want == -1 && PyErr_Occurred()) return nullptr;
    std::vector<uint8_t> b((size_t)std::max<Py_ssize_t>(want, 0));
    size_t got = 0;
    while (got < b.size()) {
        const ssize_t r = ::read(self->fd, b.data() + got, b.size() - got);
        if (r < 0) { if (errno == EINTR) continue; return os_error("read"); }
        if (r == 0) break;
        got += (size_t)r;
    }
    return PyBytes_FromStringAndSize((const char*)b.data(), (Py_ssize_t)got);
}
PyObject* File_readinto(FileObject* self, PyObject* output) {       // std::io::copy(file -> output)
    std::vector<uint8_t> all;
    if (File_Check(output)) {
        if ((PyObject*)self == output) { PyErr_SetString(PyExc_OSError, "cannot readinto self"); return nullptr; }
        if (!file_read_to_end(self, all) || !file_write_all((FileObject*)output, all.data(), all.size())) return nullptr;
        return PyLong_FromSize_t(all.size());
    }
    if (Buffer_Check(output)) {
        BufferObject* o = (BufferObject*)output;
        if (buffer_sync_view(o) < 0 || !file_read_to_end(self, all)) return nullptr;
        if (o->view) {
            const Py_ssize_t room = o->vlen - (Py_ssize_t)std::min<uint64_t>(o->pos, (uint64_t)o->vlen);
            if ((Py_ssize_t)all.size() > room) { PyErr_SetString(PyExc_OSError, "failed to write whole buffer"); return nullptr; }
            if (!all.empty()) std::memcpy(o->vptr + o->pos, all.data(), all.size());
            o->pos += all.size();
        } else owned_write(o, all.data(), all.size());
        return PyLong_FromSize_t(all.size());
    }
    Bytes out;
    if (!get_bytes(output, out)) return nullptr;
    size_t got = 0;
    while (got < (size_t)out.len) {
        const ssize_t r = ::read(self->fd, out.ptr + got, (size_t)out.len - got);
        if (r < 0) { if (errno == EINTR) continue; return os_error("read"); }
        if (r == 0) break;
        got += (size_t)r;
    }
    if (got == (size_t)out.len) {         // more data left than the output can take -> write_all fails (WriteZero)
        uint8_t probe;
        const ssize_t r = ::read(self->fd, &probe, 1);
        if (r > 0) { (void)::lseek(self->fd, -1, SEEK_CUR); PyErr_SetString(PyExc_OSError, "failed to write whole buffer"); return nullptr; }
    }
    return PyLong_FromSize_t(got);
}
PyObject* File_seek(FileObject* self, PyObject* args, PyObject* kw) {
    static const char* kwl[] = {"position", "whence", nullptr};
    Py_ssize_t position; PyObject* wobj = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "n|O", (char**)kwl, &position, &wobj)) return nullptr;
    long whence = 0;
    if (wobj != Py_None) { whence = PyLong_AsLong(wobj); if (whence == -1 && PyErr_Occurred()) return nullptr; }
    if (whence < 0 || whence > 2) {
        PyErr_SetString(PyExc_ValueError, "whence should be one of 0: seek from start, 1: seek from current, or 2: seek from end");
        return nullptr;
    }
    const off_t r = ::lseek(self->fd, (off_t)position, whence == 0 ? SEEK_SET : whence == 1 ? SEEK_CUR : SEEK_END);
    if (r < 0) return os_error("seek");
    return PyLong_FromLongLong((long long)r);
}
PyObject* File_seekable(FileObject*, PyObject*) { Py_RETURN_TRUE; }
PyObject* File_tell(FileObject* self, PyObject*) {
    const off_t r = ::lseek(self->fd, 0, SEEK_CUR);
    if (r < 0) return os_error("seek");
    return PyLong_FromLongLong((long long)r);
}
PyObject* File_set_len(FileObject* self, PyObject* arg) {
    const size_t n = PyLong_AsSize_t(arg);
    if (n == (size_t)-1 && PyErr_Occurred()) return nullptr;
    if (::ftruncate(self->fd, (off_t)n) != 0) return os_error("set_len");
    Py_RETURN_NONE;
}
PyObject* File_truncate(FileObject* self, PyObject*) {
    if (::ftruncate(self->fd, 0) != 0) return os_error("set_len");
    Py_RETURN_NONE;
}
Py_ssize_t file_len(FileObject* self) {
    struct stat st;
    if (::fstat(self->fd, &st) != 0) { os_error("metadata"); return -1; }
    return (Py_ssize_t)st.st_size;
}
PyObject* File_len(FileObject* self, PyObject*) { const Py_ssize_t n = file_len(self); return n < 0 ? nullptr : PyLong_FromSsize_t(n); }
Py_ssize_t File_sq_len(FileObject* self) { return file_len(self); }
int File_bool(FileObject* self) { const Py_ssize_t n = file_len(self); return n < 0 ? -1 : n > 0; }
PyObject* File_repr(FileObject* self) {
    const Py_ssize_t n = file_len(self);
    if (n < 0) return nullptr;
    return PyUnicode_FromFormat("cramjam.File<path=%s, len=%zd>", self->path->c_str(), n);
}
PyMethodDef File_methods[] = {
    {"write", (PyCFunction)Guarded<File_write>::call, METH_O, "Write some bytes to the file, where input data can be anything in BytesType"},
    {"read", (PyCFunction)Guarded<File_read>::call, METH_VARARGS | METH_KEYWORDS, "Read from the file in its current position, returns bytes (n_bytes=None)"},
    {"readinto", (PyCFunction)Guarded<File_readinto>::call, METH_O, "Read from the file in its current position, into a BytesType object."},
    {"seek", (PyCFunction)Guarded<File_seek>::call, METH_VARARGS | METH_KEYWORDS, "Seek to a position within the file (position, whence=None)"},
    {"seekable", (PyCFunction)Guarded<File_seekable>::call, METH_NOARGS, "Whether the file is seekable; always True."},
    {"tell", (PyCFunction)Guarded<File_tell>::call, METH_NOARGS, "Give the current position of the file."},
    {"set_len", (PyCFunction)Guarded<File_set_len>::call, METH_O, "Set the length of the file (truncates or null-byte fills)."},
    {"truncate", (PyCFunction)Guarded<File_truncate>::call, METH_NOARGS, "Truncate the file."},
    {"len", (PyCFunction)Guarded<File_len>::call, METH_NOARGS, "Length of the file in bytes"},
    {nullptr, nullptr, 0, nullptr}};
PySequenceMethods File_as_sequence = {};
PyNumberMethods File_as_number = {};
PyTypeObject FileType = { PyVarObject_HEAD_INIT(nullptr, 0) };

// ------------------------------------------------------------------------------------------
// Streaming objects (reference src/snappy.rs:124-161, src/lz4.rs:231-294, src/lib.rs:298-394, src/io.rs:761-814).
// The reference's encoders compress as blocks fill; here input is collected and every flush()/finish() compresses what
// is pending as ONE device batch — the bytes returned by the flushes concatenate to one valid framed stream.
// ------------------------------------------------------------------------------------------
struct CompressorObject {
    PyObject_HEAD
    int codec;                   // 0 snappy, 1 lz4
    bool finished, started, content_checksum;
    bool busy;                   // a flush()/finish() of this object is running with the GIL released
    int level;
    ByteVec* pending;
    cj::Xxh32* hash;
};
// pyo3 guards &mut self with a borrow flag and raises "Already borrowed" on re-entry (src/io.rs:761-814 take &mut self);
// without it a second thread calling compress()/flush() during the GPU call would reallocate the vector being read
struct CompressorBorrow {
    CompressorObject* o; bool ok;
    explicit CompressorBorrow(CompressorObject* c) : o(c), ok(!c->busy) {
        if (ok) o->busy = true; else PyErr_SetString(PyExc_RuntimeError, "Already borrowed");
    }
    ~CompressorBorrow() { if (ok) o->busy = false; }
};

extern PyTypeObject SnappyCompressorType, Lz4CompressorType, SnappyDecompressorType, Lz4DecompressorType;

PyObject* Compressor_new_common(PyTypeObject* type, int codec) {
    CompressorObject* self = (CompressorObject*)type->tp_alloc(type, 0);
    if (!self) return nullptr;
    self->codec = codec; self->finished = false; self->started = false; self->content_checksum = true; self->level = -1; self->busy = false;
    self->pending = new ByteVec();
    self->hash = new cj::Xxh32(0);
    return (PyObject*)self;
}
PyObject* SnappyCompressor_new(PyTypeObject* t, PyObject*, PyObject*) { return Compressor_new_common(t, 0); }
PyObject* Lz4Compressor_new(PyTypeObject* t, PyObject*, PyObject*) { return Compressor_new_common(t, 1); }

int SnappyCompressor_init(CompressorObject*, PyObject* args, PyObject* kw) {
    static const char* kwl[] = {nullptr};
    return PyArg_ParseTupleAndKeywords(args, kw, "", (char**)kwl) ? 0 : -1;
}
int Lz4Compressor_init(CompressorObject* self, PyObject* args, PyObject* kw) {       // src/lz4.rs:240-262
    static const char* kwl[] = {"level", "content_checksum", "block_linked", nullptr};
    PyObject *lvl = Py_None, *cs = Py_None, *bl = Py_None;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "|OOO", (char**)kwl, &lvl, &cs, &bl)) return -1;
    self->level = opt_int(lvl, -1);
    if (self->level == -2) return -1;
    if (cs != Py_None) { int t = PyObject_IsTrue(cs); if (t < 0) return -1; self->content_checksum = t != 0; }
    if (bl != Py_None && PyObject_IsTrue(bl) < 0) return -1;     // accepted; blocks are always independent here
    return 0;
}
void Compressor_dealloc(CompressorObject* self) {
    delete self->pending; delete self->hash;
    Py_TYPE(self)->tp_free((PyObject*)self);
}

PyObject* Compressor_compress(CompressorObject* self, PyObject* input) {               // src/io.rs:761-771
    if (self->finished) {
        PyErr_SetString(CompressionError, "Compressor looks to have been consumed via `finish()`. please create a new compressor instance.");
        return nullptr;
    }
    CompressorBorrow borrow(self);
    if (!borrow.ok) return nullptr;
    Bytes in;
    if (!get_bytes(input, in)) return nullptr;
    self->pending->insert(self->pending->end(), in.ptr, in.ptr + in.len);
    return PyLong_FromSsize_t(in.len);
}

// compress what is pending and append it to `out`; 0 or a CJ_E_* code
int64_t compressor_emit(CompressorObject* self, ByteVec& out, bool finish) {
    ByteVec& pend = *self->pending;
    if (self->codec == 0) {
        // snap write::FrameEncoder: the stream identifier goes out with the first chunk; nothing at all for no input
        if (!pend.empty()) {
            ByteVec tmp(cj_snappy_frame_max_compress_len(pend.size()));
            const int64_t r = cj_snappy_frame_compress(pend.data(), pend.size(), tmp.data(), tmp.size());
            if (r < 0) return r;
            const size_t skip = self->started ? 10 : 0;
            out.insert(out.end(), tmp.begin() + (long)skip, tmp.begin() + (long)r);
            self->started = true;
        }
    } else {
        if (!self->started) {                  // LZ4F_compressBegin: the header exists before any data
            uint8_t hdr[7] = { 0x04, 0x22, 0x4D, 0x18, (uint8_t)(0x60 | (self->content_checksum ? 0x04 : 0)), 0x40, 0 };
            hdr[6] = (uint8_t)(cj::xxh32(hdr + 4, 2, 0) >> 8);
            out.insert(out.end(), hdr, hdr + 7);
            self->started = true;
        }
        if (!pend.empty()) {
            if (self->content_checksum) self->hash->update(pend.data(), pend.size());
            ByteVec tmp(pend.size() + 4 * ((pend.size() + 65535) / 65536));
            const int64_t r = cj_lz4_frame_compress_blocks(pend.data(), pend.size(), tmp.data(), tmp.size());
            if (r < 0) return r;
            out.insert(out.end(), tmp.begin(), tmp.begin() + (long)r);
        }
        if (finish) {
            uint8_t tail[8] = {0};
            const uint32_t d = self->hash->digest();
            std::memcpy(tail + 4, &d, 4);
            out.insert(out.end(), tail, tail + (self->content_checksum ? 8 : 4));
        }
    }
    pend.clear();
    return 0;
}

PyObject* Compressor_flush(CompressorObject* self, PyObject*) {                        // src/io.rs:796-814
    CompressorBorrow borrow(self);
    if (!borrow.ok) return nullptr;
    ByteVec out;
    if (!self->finished) {
        int64_t r;
        Py_BEGIN_ALLOW_THREADS
        r = compressor_emit(self, out, false);
        Py_END_ALLOW_THREADS
        if (r < 0) return raise_code(CompressionError, r);
    }
    return buffer_from_vec(std::move(out));
}

PyObject* Compressor_finish(CompressorObject* self, PyObject*) {                       // src/io.rs:773-794
    CompressorBorrow borrow(self);
    if (!borrow.ok) return nullptr;
    ByteVec out;
    if (!self->finished) {
        int64_t r;
        Py_BEGIN_ALLOW_THREADS
        r = compressor_emit(self, out, true);
        Py_END_ALLOW_THREADS
        if (r < 0) return raise_code(CompressionError, r);
        self->finished = true;
    }
    return buffer_from_vec(std::move(out));
}

PyMethodDef Compressor_methods[] = {
    {"compress", (PyCFunction)Guarded<Compressor_compress>::call, METH_O, "Compress input into the current compressor's stream."},
    {"flush", (PyCFunction)Guarded<Compressor_flush>::call, METH_NOARGS, "Flush and return current compressed stream"},
    {"finish", (PyCFunction)Guarded<Compressor_finish>::call, METH_NOARGS, "Consume the current compressor state and return the compressed stream"},
    {nullptr, nullptr, 0, nullptr}};

struct DecompressorObject {                                                            // src/lib.rs:298-394
    PyObject_HEAD
    int codec;
    bool finished;
    ByteVec* inner;
};

PyObject* Decompressor_new_common(PyTypeObject* type, int codec) {
    DecompressorObject* self = (DecompressorObject*)type->tp_alloc(type, 0);
    if (!self) return nullptr;
    self->codec = codec; self->finished = false; self->inner = new ByteVec();
    return (PyObject*)self;
}
PyObject* SnappyDecompressor_new(PyTypeObject* t, PyObject*, PyObject*) { return Decompressor_new_common(t, 0); }
PyObject* Lz4Decompressor_new(PyTypeObject* t, PyObject*, PyObject*) { return Decompressor_new_common(t, 1); }
void Decompressor_dealloc(DecompressorObject* self) { delete self->inner; Py_TYPE(self)->tp_free((PyObject*)self); }

PyObject* decompressor_gone() {
    PyErr_SetString(DecompressionError, "Appears `finish()` was called on this instance");
    return nullptr;
}

PyObject* Decompressor_decompress(DecompressorObject* self, PyObject* input) {
    if (self->finished) return decompressor_gone();
    Bytes in;
    if (!get_bytes(input, in)) return nullptr;
    const Framed& fc = self->codec == 0 ? kSnappyFramed : kLz4Framed;
    int64_t r;
    ByteVec tmp;
    Py_BEGIN_ALLOW_THREADS
    r = fc.need(in, false);
    if (r >= 0) {
        tmp = make_result((size_t)r);
        r = fc.run(in, tmp.data(), tmp.size(), false, -1);
    }
    Py_END_ALLOW_THREADS
    if (r < 0) return raise_code(DecompressionError, r);
    self->inner->insert(self->inner->end(), tmp.begin(), tmp.begin() + (long)r);
    return PyLong_FromLongLong(r);
}
PyObject* Decompressor_flush(DecompressorObject* self, PyObject*) {
    if (self->finished) return decompressor_gone();
    ByteVec out;
    out.swap(*self->inner);
    return buffer_from_vec(std::move(out));
}
PyObject* Decompressor_finish(DecompressorObject* self, PyObject*) {
    if (self->finished) return decompressor_gone();
    self->finished = true;
    ByteVec out;
    out.swap(*self->inner);
    return buffer_from_vec(std::move(out));
}
PyObject* Decompressor_len(DecompressorObject* self, PyObject*) { return PyLong_FromSize_t(self->finished ? 0 : self->inner->size()); }
Py_ssize_t Decompressor_sq_len(DecompressorObject* self) { return self->finished ? 0 : (Py_ssize_t)self->inner->size(); }
int Decompressor_contains(DecompressorObject* self, PyObject* x) {
    Bytes b;
    if (!get_bytes(x, b)) return -1;
    if (self->finished) return 0;
    const ByteVec& v = *self->inner;
    if (b.len == 0) return 1;
    return std::search(v.begin(), v.end(), b.ptr, b.ptr + b.len) != v.end();
}
int Decompressor_bool(DecompressorObject* self) { return !self->finished && !self->inner->empty(); }
PyObject* Decompressor_repr(DecompressorObject* self) { return PyUnicode_FromFormat("Decompressor<len=%zu>", self->finished ? (size_t)0 : self->inner->size()); }

PyMethodDef Decompressor_methods[] = {
    {"decompress", (PyCFunction)Guarded<Decompressor_decompress>::call, METH_O, "Decompress this input into the inner buffer."},
    {"flush", (PyCFunction)Guarded<Decompressor_flush>::call, METH_NOARGS, "Flush and return current decompressed stream."},
    {"finish", (PyCFunction)Guarded<Decompressor_finish>::call, METH_NOARGS, "Consume the current Decompressor state and return the decompressed stream"},
    {"len", (PyCFunction)Guarded<Decompressor_len>::call, METH_NOARGS, "Length of internal buffer containing decompressed data."},
    {nullptr, nullptr, 0, nullptr}};
PySequenceMethods Decompressor_as_sequence = {};
PyNumberMethods Decompressor_as_number = {};

PyTypeObject SnappyCompressorType = { PyVarObject_HEAD_INIT(nullptr, 0) };
PyTypeObject Lz4CompressorType = { PyVarObject_HEAD_INIT(nullptr, 0) };
PyTypeObject SnappyDecompressorType = { PyVarObject_HEAD_INIT(nullptr, 0) };
PyTypeObject Lz4DecompressorType = { PyVarObject_HEAD_INIT(nullptr, 0) };

bool ready_stream_types() {
    struct { PyTypeObject* t; const char* name; newfunc nw; initproc in; } cs[] = {
        { &SnappyCompressorType, "cramjam_amd.snappy.Compressor", SnappyCompressor_new, (initproc)SnappyCompressor_init },
        { &Lz4CompressorType, "cramjam_amd.lz4.Compressor", Lz4Compressor_new, (initproc)Lz4Compressor_init } };
    for (auto& c : cs) {
        c.t->tp_name = c.name; c.t->tp_basicsize = sizeof(CompressorObject); c.t->tp_flags = Py_TPFLAGS_DEFAULT;
        c.t->tp_doc = "Compressor object for streaming compression"; c.t->tp_new = c.nw; c.t->tp_init = c.in;
        c.t->tp_dealloc = (destructor)Compressor_dealloc; c.t->tp_methods = Compressor_methods;
        if (PyType_Ready(c.t) < 0) return false;
    }
    Decompressor_as_sequence.sq_length = (lenfunc)Decompressor_sq_len;
    Decompressor_as_sequence.sq_contains = (objobjproc)Decompressor_contains;
    Decompressor_as_number.nb_bool = (inquiry)Decompressor_bool;
    struct { PyTypeObject* t; const char* name; newfunc nw; } ds[] = {
        { &SnappyDecompressorType, "cramjam_amd.snappy.Decompressor", SnappyDecompressor_new },
        { &Lz4DecompressorType, "cramjam_amd.lz4.Decompressor", Lz4Decompressor_new } };
    for (auto& d : ds) {
        d.t->tp_name = d.name; d.t->tp_basicsize = sizeof(DecompressorObject); d.t->tp_flags = Py_TPFLAGS_DEFAULT;
        d.t->tp_doc = "Decompressor object for streaming decompression"; d.t->tp_new = d.nw;
        d.t->tp_dealloc = (destructor)Decompressor_dealloc; d.t->tp_methods = Decompressor_methods;
        d.t->tp_as_sequence = &Decompressor_as_sequence; d.t->tp_as_number = &Decompressor_as_number;
        d.t->tp_repr = (reprfunc)Decompressor_repr;
        if (PyType_Ready(d.t) < 0) return false;
    }
    return true;
}

PyMethodDef lz4_methods[] = {
    {"compress", (PyCFunction)Guarded<lz4_compress>::call, METH_VARARGS | METH_KEYWORDS, "LZ4 (frame) compression (data, level=None, output_len=None)"},
    {"decompress", (PyCFunction)Guarded<lz4_decompress>::call, METH_VARARGS | METH_KEYWORDS, "LZ4 (frame) decompression (data, output_len=None)"},
    {"compress_into", (PyCFunction)Guarded<lz4_compress_into>::call, METH_VARARGS | METH_KEYWORDS, "Compress (frame) directly into an output buffer (input, output, level=None)"},
    {"decompress_into", (PyCFunction)Guarded<lz4_decompress_into>::call, METH_VARARGS | METH_KEYWORDS, "Decompress (frame) directly into an output buffer (input, output)"},
    {"decompress_block", (PyCFunction)Guarded<lz4_decompress_block>::call, METH_VARARGS | METH_KEYWORDS, "LZ4 block decompression (data, output_len=None)"},
    {"compress_block", (PyCFunction)Guarded<lz4_compress_block>::call, METH_VARARGS | METH_KEYWORDS, "LZ4 block compression (data, output_len=None, mode=None, acceleration=None, compression=None, store_size=None)"},
    {"decompress_block_into", (PyCFunction)Guarded<lz4_decompress_block_into>::call, METH_VARARGS | METH_KEYWORDS, "LZ4 block decompression into a pre-allocated buffer (input, output, output_len=None)"},
    {"compress_block_into", (PyCFunction)Guarded<lz4_compress_block_into>::call, METH_VARARGS | METH_KEYWORDS, "LZ4 block compression into a pre-allocated buffer"},
    {"compress_block_bound", (PyCFunction)Guarded<lz4_compress_block_bound>::call, METH_O, "Size of a buffer guaranteed to hold the block-compressed result"},
    {nullptr, nullptr, 0, nullptr}};

PyMethodDef snappy_methods[] = {
    {"compress", (PyCFunction)Guarded<snappy_compress>::call, METH_VARARGS | METH_KEYWORDS, "Snappy (framed) compression (data, output_len=None)"},
    {"decompress", (PyCFunction)Guarded<snappy_decompress>::call, METH_VARARGS | METH_KEYWORDS, "Snappy (framed) decompression (data, output_len=None)"},
    {"compress_into", (PyCFunction)Guarded<snappy_compress_into>::call, METH_VARARGS | METH_KEYWORDS, "Compress (framed) directly into an output buffer"},
    {"decompress_into", (PyCFunction)Guarded<snappy_decompress_into>::call, METH_VARARGS | METH_KEYWORDS, "Decompress (framed) directly into an output buffer"},
    {"decompress_raw", (PyCFunction)Guarded<snappy_decompress_raw>::call, METH_VARARGS | METH_KEYWORDS, "Snappy raw decompression (data, output_len=None)"},
    {"compress_raw", (PyCFunction)Guarded<snappy_compress_raw>::call, METH_VARARGS | METH_KEYWORDS, "Snappy raw compression (data, output_len=None)"},
    {"compress_raw_into", (PyCFunction)Guarded<snappy_compress_raw_into>::call, METH_VARARGS | METH_KEYWORDS, "Compress raw format directly into an output buffer"},
    {"decompress_raw_into", (PyCFunction)Guarded<snappy_decompress_raw_into>::call, METH_VARARGS | METH_KEYWORDS, "Decompress raw format directly into an output buffer"},
    {"compress_raw_max_len", (PyCFunction)Guarded<snappy_compress_raw_max_len>::call, METH_O, "Max compressed length for snappy raw compression"},
    {"decompress_raw_len", (PyCFunction)Guarded<snappy_decompress_raw_len>::call, METH_O, "Decompressed length of the given raw data"},
    {nullptr, nullptr, 0, nullptr}};

// ---- batch extension: many independent chunks per call, host buffers in, `bytes` out (cramjam_amd/batch.py) ----------------
// batch_host(engine_handle, codec, op, flags, inputs, out_caps) -> (results, outputs).  The reference has one call per buffer
// (src/lz4.rs:78-131, src/snappy.rs:52-78); this is the same borrowing (buffer protocol, no copy of the inputs) for a list of
// them, and the outputs are `bytes` objects the engine scatters INTO — the ctypes marshalling this replaces copied every
// input and every output once more and spent 130 ms of a 160 ms call on 16 384 chunks in Python objects.
typedef int (*batch_host_fn)(cj_engine*, cj_codec, cj_op, uint32_t, size_t, const uint8_t* const*, const size_t*, uint8_t* const*, const size_t*, int64_t*);
PyObject* root_batch_host(PyObject*, PyObject* args) {
    unsigned long long handle, fn_addr = 0; int codec, op; unsigned int flags; PyObject *inputs_o, *caps_o;
    if (!PyArg_ParseTuple(args, "KiiIOO|K", &handle, &codec, &op, &flags, &inputs_o, &caps_o, &fn_addr)) return nullptr;
    // (fn_addr: cj_batch_host of the library the engine handle came from — a tuning variant loaded through CJ_HIP_LIB; 0 = the one this module links)
    const batch_host_fn call = fn_addr ? (batch_host_fn)(uintptr_t)fn_addr : &cj_batch_host;
    PyObject* inputs = PySequence_Fast(inputs_o, "inputs must be a sequence of bytes-like objects");
    if (!inputs) return nullptr;
    PyObject* caps = PySequence_Fast(caps_o, "out_caps must be a sequence of integers");
    if (!caps) { Py_DECREF(inputs); return nullptr; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(inputs);
    if (PySequence_Fast_GET_SIZE(caps) != n) { Py_DECREF(inputs); Py_DECREF(caps); PyErr_SetString(PyExc_ValueError, "inputs and out_caps differ in length"); return nullptr; }
    std::vector<Py_buffer> views((size_t)n);
    std::vector<const uint8_t*> in_ptrs((size_t)n);
    std::vector<size_t> in_lens((size_t)n), out_caps((size_t)n);
    std::vector<uint8_t*> out_ptrs((size_t)n);
    std::vector<int64_t> res((size_t)n);
    Py_ssize_t got = 0;
    PyObject* outs = PyList_New(n);
    bool ok = outs != nullptr;
    for (Py_ssize_t i = 0; ok && i < n; i++) {
        if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(inputs, i), &views[(size_t)i], PyBUF_CONTIG_RO) != 0) { ok = false; break; }
        got = i + 1;
        in_lens[(size_t)i] = (size_t)views[(size_t)i].len;
        in_ptrs[(size_t)i] = views[(size_t)i].len ? (const uint8_t*)views[(size_t)i].buf : nullptr;
        const size_t cap = PyLong_AsSize_t(PySequence_Fast_GET_ITEM(caps, i));
        if (cap == (size_t)-1 && PyErr_Occurred()) { ok = false; break; }
        PyObject* b = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)(cap ? cap : 1));      // (a zero capacity still gets an address)
        if (!b) { ok = false; break; }
        PyList_SET_ITEM(outs, i, b);
        out_caps[(size_t)i] = cap;
        out_ptrs[(size_t)i] = (uint8_t*)PyBytes_AS_STRING(b);
    }
    int rc = 0;
    if (ok && n > 0) {
        Py_BEGIN_ALLOW_THREADS
        rc = call((cj_engine*)(uintptr_t)handle, (cj_codec)codec, (cj_op)op, flags, (size_t)n, in_ptrs.data(), in_lens.data(), out_ptrs.data(), out_caps.data(), res.data());
        Py_END_ALLOW_THREADS
    }
    for (Py_ssize_t i = 0; i < got; i++) PyBuffer_Release(&views[(size_t)i]);
    Py_DECREF(inputs); Py_DECREF(caps);
    if (ok && rc != 0) {
        // (the HIP error text is thread-local PER LIBRARY: through a tuning variant's entry point it lies in that library, and the caller —
        //  cramjam_amd/_native.py, which holds the variant's handle — appends it; round-5 advisor)
        if (fn_addr) PyErr_Format(PyExc_RuntimeError, "cramjam_hip error %d: %s", rc, cj_strerror(rc));
        else PyErr_Format(PyExc_RuntimeError, "cramjam_hip error %d: %s (%s)", rc, cj_strerror(rc), cj_last_hip_error());
        ok = false;
    }
    PyObject* results = ok ? PyList_New(n) : nullptr;
    if (!results) ok = false;
    for (Py_ssize_t i = 0; ok && i < n; i++) {
        PyObject* r = PyLong_FromLongLong((long long)res[(size_t)i]);
        if (!r) { ok = false; break; }
        PyList_SET_ITEM(results, i, r);
        const Py_ssize_t want = res[(size_t)i] > 0 ? (Py_ssize_t)res[(size_t)i] : 0;
        PyObject* b = PyList_GET_ITEM(outs, i);
        if (PyBytes_GET_SIZE(b) != want) {                       // (the list holds the only reference: resized in place or moved)
            PyList_SET_ITEM(outs, i, nullptr);
            if (_PyBytes_Resize(&b, want) != 0) { ok = false; break; }
            PyList_SET_ITEM(outs, i, b);
        }
    }
    if (!ok) { Py_XDECREF(outs); Py_XDECREF(results); return nullptr; }
    PyObject* t = PyTuple_Pack(2, results, outs);
    Py_DECREF(results); Py_DECREF(outs);
    return t;
}

// batch_host_into(engine_handle, codec, op, flags, inputs, out_caps, out, offsets=None) -> results: the same batch into ONE writable buffer
// of the caller's (chunk i at out[offsets[i] : offsets[i] + out_caps[i]], back to back when offsets is None) — no object per output
PyObject* root_batch_host_into(PyObject*, PyObject* args) {
    unsigned long long handle, fn_addr = 0; int codec, op; unsigned int flags; PyObject *inputs_o, *caps_o, *out_o, *offs_o = Py_None;
    if (!PyArg_ParseTuple(args, "KiiIOOO|OK", &handle, &codec, &op, &flags, &inputs_o, &caps_o, &out_o, &offs_o, &fn_addr)) return nullptr;
    const batch_host_fn call = fn_addr ? (batch_host_fn)(uintptr_t)fn_addr : &cj_batch_host;
    Py_buffer ob;
    if (PyObject_GetBuffer(out_o, &ob, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return nullptr;
    PyObject* inputs = PySequence_Fast(inputs_o, "inputs must be a sequence of bytes-like objects");
    PyObject* caps = inputs ? PySequence_Fast(caps_o, "out_caps must be a sequence of integers") : nullptr;
    PyObject* offs = (caps && offs_o != Py_None) ? PySequence_Fast(offs_o, "offsets must be a sequence of integers") : nullptr;
    bool ok = inputs && caps && (offs_o == Py_None || offs);
    const Py_ssize_t n = ok ? PySequence_Fast_GET_SIZE(inputs) : 0;
    if (ok && (PySequence_Fast_GET_SIZE(caps) != n || (offs && PySequence_Fast_GET_SIZE(offs) != n))) { PyErr_SetString(PyExc_ValueError, "inputs, out_caps and offsets differ in length"); ok = false; }
    std::vector<Py_buffer> views((size_t)n);
    std::vector<const uint8_t*> in_ptrs((size_t)n);
    std::vector<size_t> in_lens((size_t)n), out_caps((size_t)n);
    std::vector<uint8_t*> out_ptrs((size_t)n);
    std::vector<int64_t> res((size_t)n);
    Py_ssize_t got = 0;
    size_t run = 0;
    for (Py_ssize_t i = 0; ok && i < n; i++) {
        if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(inputs, i), &views[(size_t)i], PyBUF_CONTIG_RO) != 0) { ok = false; break; }
        got = i + 1;
        in_lens[(size_t)i] = (size_t)views[(size_t)i].len;
        in_ptrs[(size_t)i] = views[(size_t)i].len ? (const uint8_t*)views[(size_t)i].buf : nullptr;
        const size_t cap = PyLong_AsSize_t(PySequence_Fast_GET_ITEM(caps, i));
        if (cap == (size_t)-1 && PyErr_Occurred()) { ok = false; break; }
        size_t off = run;
        if (offs) { off = PyLong_AsSize_t(PySequence_Fast_GET_ITEM(offs, i)); if (off == (size_t)-1 && PyErr_Occurred()) { ok = false; break; } }
        if (off > (size_t)ob.len || cap > (size_t)ob.len - off) { PyErr_SetString(PyExc_ValueError, "out is too small for the capacities given"); ok = false; break; }
        out_caps[(size_t)i] = cap;
        out_ptrs[(size_t)i] = (uint8_t*)ob.buf + off;
        run = off + cap;
    }
    int rc = 0;
    if (ok && n > 0) {
        Py_BEGIN_ALLOW_THREADS
        rc = call((cj_engine*)(uintptr_t)handle, (cj_codec)codec, (cj_op)op, flags, (size_t)n, in_ptrs.data(), in_lens.data(), out_ptrs.data(), out_caps.data(), res.data());
        Py_END_ALLOW_THREADS
    }
    for (Py_ssize_t i = 0; i < got; i++) PyBuffer_Release(&views[(size_t)i]);
    PyBuffer_Release(&ob);
    Py_XDECREF(inputs); Py_XDECREF(caps); Py_XDECREF(offs);
    if (ok && rc != 0) { // (the HIP error text is thread-local PER LIBRARY: through a tuning variant's entry point it lies in that library, and the caller —
        //  cramjam_amd/_native.py, which holds the variant's handle — appends it; round-5 advisor)
        if (fn_addr) PyErr_Format(PyExc_RuntimeError, "cramjam_hip error %d: %s", rc, cj_strerror(rc));
        else PyErr_Format(PyExc_RuntimeError, "cramjam_hip error %d: %s (%s)", rc, cj_strerror(rc), cj_last_hip_error()); ok = false; }
    if (!ok) return nullptr;
    PyObject* results = PyList_New(n);
    if (!results) return nullptr;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* r = PyLong_FromLongLong((long long)res[(size_t)i]);
        if (!r) { Py_DECREF(results); return nullptr; }
        PyList_SET_ITEM(results, i, r);
    }
    return results;
}

PyMethodDef root_methods[] = {
    {"batch_host_into", (PyCFunction)Guarded<root_batch_host_into>::call, METH_VARARGS, "batch_host_into(engine_handle, codec, op, flags, inputs, out_caps, out, offsets=None) -> results"},
    {"batch_host", (PyCFunction)Guarded<root_batch_host>::call, METH_VARARGS, "batch_host(engine_handle, codec, op, flags, inputs, out_caps) -> (results, outputs)"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef lz4_def = {PyModuleDef_HEAD_INIT, "cramjam_amd.lz4", "LZ4 block de/compression on MI355X", -1, lz4_methods};
PyModuleDef snappy_def = {PyModuleDef_HEAD_INIT, "cramjam_amd.snappy", "Snappy raw de/compression on MI355X", -1, snappy_methods};
PyModuleDef root_def = {PyModuleDef_HEAD_INIT, "cramjam_amd._cramjam", "native host layer of cramjam_amd", -1, root_methods};

}  // namespace

PyMODINIT_FUNC PyInit__cramjam(void) {
    BufferType.tp_name = "cramjam_amd.Buffer";
    BufferType.tp_basicsize = sizeof(BufferObject);
    BufferType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE;
    BufferType.tp_doc = "A native file-like in-memory buffer (cramjam.Buffer)";
    BufferType.tp_new = Buffer_new;
    BufferType.tp_init = (initproc)Buffer_init;
    BufferType.tp_dealloc = (destructor)Buffer_dealloc;
    BufferType.tp_methods = Buffer_methods;
    BufferType.tp_as_sequence = &Buffer_as_sequence;
    Buffer_as_number.nb_bool = (inquiry)Buffer_bool;
    BufferType.tp_as_number = &Buffer_as_number;
    BufferType.tp_as_buffer = &Buffer_as_buffer;
    BufferType.tp_repr = (reprfunc)Buffer_repr;
    BufferType.tp_richcompare = Buffer_richcompare;
    if (PyType_Ready(&BufferType) < 0) return nullptr;

    PyObject* m = PyModule_Create(&root_def);
    if (!m) return nullptr;
    CompressionError = PyErr_NewException("cramjam_amd.CompressionError", nullptr, nullptr);
    DecompressionError = PyErr_NewException("cramjam_amd.DecompressionError", nullptr, nullptr);
    Py_INCREF(&BufferType);
    PyModule_AddObject(m, "Buffer", (PyObject*)&BufferType);
    Py_INCREF(CompressionError); PyModule_AddObject(m, "CompressionError", CompressionError);
    Py_INCREF(DecompressionError); PyModule_AddObject(m, "DecompressionError", DecompressionError);
    PyObject* lz4 = PyModule_Create(&lz4_def);
    PyObject* snappy = PyModule_Create(&snappy_def);
    if (!lz4 || !snappy) return nullptr;
    FileType.tp_name = "cramjam_amd.File";
    FileType.tp_basicsize = sizeof(FileObject);
    FileType.tp_flags = Py_TPFLAGS_DEFAULT;
    FileType.tp_doc = "A native file object (cramjam.File)";
    FileType.tp_new = File_new;
    FileType.tp_init = (initproc)File_init;
    FileType.tp_dealloc = (destructor)File_dealloc;
    FileType.tp_methods = File_methods;
    File_as_sequence.sq_length = (lenfunc)File_sq_len;
    File_as_number.nb_bool = (inquiry)File_bool;
    FileType.tp_as_sequence = &File_as_sequence;
    FileType.tp_as_number = &File_as_number;
    FileType.tp_repr = (reprfunc)File_repr;
    if (PyType_Ready(&FileType) < 0) return nullptr;
    Py_INCREF(&FileType);
    PyModule_AddObject(m, "File", (PyObject*)&FileType);
    if (!ready_stream_types()) return nullptr;
    Py_INCREF(&Lz4CompressorType); PyModule_AddObject(lz4, "Compressor", (PyObject*)&Lz4CompressorType);
    Py_INCREF(&Lz4DecompressorType); PyModule_AddObject(lz4, "Decompressor", (PyObject*)&Lz4DecompressorType);
    Py_INCREF(&SnappyCompressorType); PyModule_AddObject(snappy, "Compressor", (PyObject*)&SnappyCompressorType);
    Py_INCREF(&SnappyDecompressorType); PyModule_AddObject(snappy, "Decompressor", (PyObject*)&SnappyDecompressorType);
    PyModule_AddObject(m, "lz4", lz4);
    PyModule_AddObject(m, "snappy", snappy);
    PyModule_AddIntConstant(m, "abi_version", cj_abi_version());
    return m;
}
