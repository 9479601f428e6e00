// _nbls_host: CPython helpers for the host side of narrow_band_least_squares().
//
// The reference returns, for LTS runs, a dictionary with one entry per (band, window) that dropped
// an element pair: key '<band:02d>_' + str(window time) -> 1-based element numbers of both members of
// every zero-weight pair (lts_array's `stdict`, merged at narrow_band_least_squares.py:114-124).  At
// cfg-3 that is ~5*10^4 string keys and small arrays per call; building them with per-window Python
// calls costs more than the whole GPU pass.  These two functions build the same objects in one C++ pass:
//
//   time_keys(t (B, VL) float64, nwin (B,) int64, prefixes list[str] | None) -> flat list[str]
//       prefix[b] + repr(float(t[b, w])) for w < nwin[b], bands in order.  repr() is Python's
//       float repr (shortest round-trip digits, fixed notation for 1e-4 <= |x| < 1e16, else exponent
//       form), which is also the text str(numpy.float64) gives.  Needs no GPU result: the band loop calls
//       it while the pass is running.
//   build_stdict(mask (B, VL, MB) uint8, nwin (B,) int64, pair_idx (P, 2) int32, nchans, keys flat list,
//                into dict | None, k0, cache | None, u0 = -1, u1 = -1)
//       u0 <= u1 given: only the units [u0, u1) of the flattened (band, window) order are entered — the unit batches of
//       a streamed pass arrive one after the other, each a range that may start and end inside a band; 'size' goes in
//       with the range that holds the last window of the first band.
//       -> dict (``into`` updated in place when given: the band groups of a pipelined call arrive one after the
//       other; k0 = index in ``keys`` of this mask's first window), entries in (band, window) order, 'size' right after the first band's entries (the
//       insertion order of the reference's merge loop).  Windows that dropped the SAME set of pairs
//       share ONE read-only array object (the reference makes a fresh array per window; the values are
//       equal, an in-place write raises instead of aliasing) — creating ~5*10^4 tiny arrays costs as
//       much host time as the GPU pass.
//
// engine.py holds pure-Python equivalents (used when this module is not built; tests compare the two).
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// Python's repr(float) into out (>= 32 bytes); returns the length.
int py_float_repr(double x, char* out) {
    if (x != x) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(x)) {
        if (x < 0) { memcpy(out, "-inf", 4); return 4; }
        memcpy(out, "inf", 3); return 3;
    }
    if (x == 0.0) {
        if (std::signbit(x)) { memcpy(out, "-0.0", 4); return 4; }
        memcpy(out, "0.0", 3); return 3;
    }
    char buf[48];
    const auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest round trip
    const char* p = buf;
    const char* end = r.ptr;
    char* o = out;
    if (*p == '-') { *o++ = '-'; ++p; }
    char digits[24];
    int nd = 0;
    while (p < end && *p != 'e') {
        if (*p != '.') digits[nd++] = *p;
        ++p;
    }
    ++p;                                    // 'e'
    int esign = 1;
    if (*p == '+') ++p; else if (*p == '-') { esign = -1; ++p; }
    int e10 = 0;
    while (p < end) e10 = e10 * 10 + (*p++ - '0');
    const int decpt = esign * e10 + 1;      // position of the decimal point relative to the digit string
    if (decpt > 16 || decpt < -3) {         // exponent form: d[.ddd]e+XX (at least two exponent digits)
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, digits + 1, nd - 1); o += nd - 1; }
        *o++ = 'e';
        int ex = decpt - 1;
        if (ex < 0) { *o++ = '-'; ex = -ex; } else *o++ = '+';
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + ex % 10); ex /= 10; } while (ex);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *o++ = eb[--ne];
    } else if (decpt <= 0) {                // 0.000ddd
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -decpt; ++i) *o++ = '0';
        memcpy(o, digits, nd); o += nd;
    } else if (decpt >= nd) {               // ddd000.0
        memcpy(o, digits, nd); o += nd;
        for (int i = nd; i < decpt; ++i) *o++ = '0';
        *o++ = '.'; *o++ = '0';
    } else {                                // dd.ddd
        memcpy(o, digits, decpt); o += decpt;
        *o++ = '.';
        memcpy(o, digits + decpt, nd - decpt); o += nd - decpt;
    }
    return (int)(o - out);
}

PyArrayObject* as_array(PyObject* obj, int typenum, int ndim, const char* what) {
    PyArrayObject* a = (PyArrayObject*)PyArray_FROM_OTF(obj, typenum, NPY_ARRAY_IN_ARRAY);
    if (!a) return nullptr;
    if (PyArray_NDIM(a) != ndim) {
        PyErr_Format(PyExc_ValueError, "%s must have %d dimensions", what, ndim);
        Py_DECREF(a);
        return nullptr;
    }
    return a;
}

PyObject* float_repr(PyObject*, PyObject* arg) {
    const double x = PyFloat_AsDouble(arg);
    if (x == -1.0 && PyErr_Occurred()) return nullptr;
    char buf[40];
    const int n = py_float_repr(x, buf);
    return PyUnicode_FromStringAndSize(buf, n);
}

PyObject* time_keys(PyObject*, PyObject* args) {
    PyObject *t_obj, *nwin_obj, *prefixes;
    if (!PyArg_ParseTuple(args, "OOO", &t_obj, &nwin_obj, &prefixes)) return nullptr;
    PyArrayObject* t = as_array(t_obj, NPY_FLOAT64, 2, "t");
    if (!t) return nullptr;
    PyArrayObject* nw = as_array(nwin_obj, NPY_INT64, 1, "nwin");
    if (!nw) { Py_DECREF(t); return nullptr; }
    const npy_intp B = PyArray_DIM(t, 0), VL = PyArray_DIM(t, 1);
    PyObject* out = nullptr;
    std::vector<std::pair<const char*, Py_ssize_t>> pre;
    bool ok = PyArray_DIM(nw, 0) == B;
    if (!ok) PyErr_SetString(PyExc_ValueError, "nwin must have one entry per band");
    if (ok && prefixes != Py_None) {
        if (!PyList_Check(prefixes) || PyList_GET_SIZE(prefixes) != B) {
            PyErr_SetString(PyExc_ValueError, "prefixes must be None or a list with one string per band");
            ok = false;
        }
        for (npy_intp b = 0; ok && b < B; ++b) {
            Py_ssize_t n = 0;
            const char* s = PyUnicode_AsUTF8AndSize(PyList_GET_ITEM(prefixes, b), &n);
            if (!s) ok = false; else pre.emplace_back(s, n);
        }
    }
    if (ok) {
        const int64_t* nwin = (const int64_t*)PyArray_DATA(nw);
        const double* tt = (const double*)PyArray_DATA(t);
        Py_ssize_t total = 0;
        for (npy_intp b = 0; b < B; ++b) {
            if (nwin[b] < 0 || nwin[b] > VL) { PyErr_SetString(PyExc_ValueError, "nwin out of range"); ok = false; break; }
            total += (Py_ssize_t)nwin[b];
        }
        if (ok) out = PyList_New(total);
        if (out) {
            Py_ssize_t k = 0;
            std::vector<char> buf(64);
            for (npy_intp b = 0; b < B && out; ++b) {
                const Py_ssize_t pl = pre.empty() ? 0 : pre[b].second;
                if ((size_t)pl + 40 > buf.size()) buf.resize(pl + 64);
                if (pl) memcpy(buf.data(), pre[b].first, pl);
                for (int64_t w = 0; w < nwin[b]; ++w) {
                    const int n = py_float_repr(tt[b * VL + w], buf.data() + pl);
                    PyObject* s = PyUnicode_FromStringAndSize(buf.data(), pl + n);
                    if (!s) { Py_CLEAR(out); break; }
                    (void)PyObject_Hash(s);          // cached in the string: the dictionary insert later skips it
                    PyList_SET_ITEM(out, k++, s);
                }
            }
        }
    }
    Py_DECREF(t);
    Py_DECREF(nw);
    return out;
}

// time_key_text(t (B, VL) float64, nwin (B,) int64, prefixes list[str] | None, nthreads) -> (text (K, 40) uint8, length (K,) uint8)
// The key text of time_keys() WITHOUT the string objects: formatted with the GIL released, on `nthreads` threads.
// build_stdict() accepts the pair in place of the key list and creates a string only for the windows that get an
// entry (two thirds of them at the benchmark's configuration), so the main thread spends ~1 ms on key objects
// instead of 4 ms.
constexpr int KEYW = 40;
PyObject* time_key_text(PyObject*, PyObject* args) {
    PyObject *t_obj, *nwin_obj, *prefixes;
    int nthreads = 1;
    if (!PyArg_ParseTuple(args, "OOO|i", &t_obj, &nwin_obj, &prefixes, &nthreads)) return nullptr;
    PyArrayObject* t = as_array(t_obj, NPY_FLOAT64, 2, "t");
    if (!t) return nullptr;
    PyArrayObject* nw = as_array(nwin_obj, NPY_INT64, 1, "nwin");
    if (!nw) { Py_DECREF(t); return nullptr; }
    const npy_intp B = PyArray_DIM(t, 0), VL = PyArray_DIM(t, 1);
    std::vector<std::string> pre;
    bool ok = PyArray_DIM(nw, 0) == B;
    if (!ok) PyErr_SetString(PyExc_ValueError, "nwin must have one entry per band");
    if (ok && prefixes != Py_None) {
        if (!PyList_Check(prefixes) || PyList_GET_SIZE(prefixes) != B) {
            PyErr_SetString(PyExc_ValueError, "prefixes must be None or a list with one string per band");
            ok = false;
        }
        for (npy_intp b = 0; ok && b < B; ++b) {
            Py_ssize_t n = 0;
            const char* s = PyUnicode_AsUTF8AndSize(PyList_GET_ITEM(prefixes, b), &n);
            if (!s) ok = false;
            else if (n > KEYW - 26) { PyErr_SetString(PyExc_ValueError, "key prefix too long for the text form"); ok = false; }
            else pre.emplace_back(s, (size_t)n);
        }
    }
    const int64_t* nwin = (const int64_t*)PyArray_DATA(nw);
    const double* tt = (const double*)PyArray_DATA(t);
    std::vector<Py_ssize_t> first((size_t)B + 1, 0);
    for (npy_intp b = 0; ok && b < B; ++b) {
        if (nwin[b] < 0 || nwin[b] > VL) { PyErr_SetString(PyExc_ValueError, "nwin out of range"); ok = false; break; }
        first[b + 1] = first[b] + (Py_ssize_t)nwin[b];
    }
    PyObject *text = nullptr, *len = nullptr, *out = nullptr;
    if (ok) {
        npy_intp d2[2] = {(npy_intp)first[B], KEYW}, d1[1] = {(npy_intp)first[B]};
        text = PyArray_SimpleNew(2, d2, NPY_UINT8);
        len = PyArray_SimpleNew(1, d1, NPY_UINT8);
    }
    if (text && len) {
        char* tx = (char*)PyArray_DATA((PyArrayObject*)text);
        uint8_t* ln = (uint8_t*)PyArray_DATA((PyArrayObject*)len);
        const bool has_pre = !pre.empty();
        auto work = [&](npy_intp b0, npy_intp b1) {
            for (npy_intp b = b0; b < b1; ++b) {
                const size_t pl = has_pre ? pre[b].size() : 0;
                for (int64_t w = 0; w < nwin[b]; ++w) {
                    char* o = tx + (size_t)(first[b] + w) * KEYW;
                    if (pl) memcpy(o, pre[b].data(), pl);
                    ln[first[b] + w] = (uint8_t)(pl + (size_t)py_float_repr(tt[b * VL + w], o + pl));
                }
            }
        };
        Py_BEGIN_ALLOW_THREADS
        int nt = nthreads < 1 ? 1 : (nthreads > 16 ? 16 : nthreads);
        if ((npy_intp)nt > B) nt = B > 0 ? (int)B : 1;
        if (nt <= 1) work(0, B);
        else {
            std::vector<std::thread> th;
            for (int i = 1; i < nt; ++i) th.emplace_back(work, B * i / nt, B * (i + 1) / nt);
            work(0, B / nt);
            for (auto& x : th) x.join();
        }
        Py_END_ALLOW_THREADS
        out = PyTuple_Pack(2, text, len);
    }
    Py_XDECREF(text);
    Py_XDECREF(len);
    Py_DECREF(t);
    Py_DECREF(nw);
    return out;
}

// Shared value arrays of the dropped-pair patterns, kept ACROSS the build_stdict() calls of one pipelined call (one
// call per band group): a pattern seen in group 0 is not built again for group 1.  Lives in a capsule the caller holds
// for the duration of its call; tied to one pair table (size and content hash), reset when another one shows up.
struct PatternCache {
    std::unordered_map<std::string, PyObject*> patterns;
    std::unordered_map<uint64_t, PyObject*> patterns64;
    npy_intp P = -1;
    uint64_t pair_hash = 0;
    void clear() {
        for (auto& kv : patterns) Py_DECREF(kv.second);
        for (auto& kv : patterns64) Py_DECREF(kv.second);
        patterns.clear();
        patterns64.clear();
    }
};
const char* const CACHE_NAME = "narrow_band_least_squares_amd.pattern_cache";
void cache_free(PyObject* cap) {
    PatternCache* c = (PatternCache*)PyCapsule_GetPointer(cap, CACHE_NAME);
    if (c) { c->clear(); delete c; }
}
PyObject* new_pattern_cache(PyObject*, PyObject*) {
    PatternCache* c = new PatternCache();
    PyObject* cap = PyCapsule_New(c, CACHE_NAME, cache_free);
    if (!cap) delete c;
    return cap;
}

PyObject* build_stdict(PyObject*, PyObject* args) {
    PyObject *mask_obj, *nwin_obj, *pair_obj, *keys, *into = Py_None, *cache_obj = Py_None;
    long nchans;
    Py_ssize_t k0 = 0, ru0 = -1, ru1 = -1;
    if (!PyArg_ParseTuple(args, "OOOlO|OnOnn", &mask_obj, &nwin_obj, &pair_obj, &nchans, &keys, &into, &k0, &cache_obj, &ru0, &ru1)) return nullptr;
    if (into != Py_None && !PyDict_Check(into)) { PyErr_SetString(PyExc_TypeError, "into must be a dict or None"); return nullptr; }
    PatternCache* cache = nullptr;
    if (cache_obj != Py_None) {
        cache = PyCapsule_CheckExact(cache_obj) ? (PatternCache*)PyCapsule_GetPointer(cache_obj, CACHE_NAME) : nullptr;
        if (!cache) { PyErr_SetString(PyExc_TypeError, "cache must be None or a new_pattern_cache() object"); return nullptr; }
    }
    // keys: the flat list of time_keys(), or the (text, length) pair of time_key_text()
    PyArrayObject *ktext = nullptr, *klen = nullptr;
    if (PyTuple_Check(keys) && PyTuple_GET_SIZE(keys) == 2) {
        ktext = as_array(PyTuple_GET_ITEM(keys, 0), NPY_UINT8, 2, "key text");
        if (!ktext) return nullptr;
        klen = as_array(PyTuple_GET_ITEM(keys, 1), NPY_UINT8, 1, "key lengths");
        if (!klen || PyArray_DIM(ktext, 1) != KEYW || PyArray_DIM(klen, 0) != PyArray_DIM(ktext, 0)) {
            if (klen) PyErr_SetString(PyExc_ValueError, "build_stdict: malformed key text");
            Py_DECREF(ktext);
            Py_XDECREF(klen);
            return nullptr;
        }
    } else if (!PyList_Check(keys)) {
        PyErr_SetString(PyExc_TypeError, "keys must be a flat list of strings or the pair returned by time_key_text()");
        return nullptr;
    }
    const Py_ssize_t nkeys = ktext ? (Py_ssize_t)PyArray_DIM(ktext, 0) : PyList_GET_SIZE(keys);
    PyArrayObject* mask = as_array(mask_obj, NPY_UINT8, 3, "mask");
    if (!mask) { Py_XDECREF(ktext); Py_XDECREF(klen); return nullptr; }
    PyArrayObject* nw = as_array(nwin_obj, NPY_INT64, 1, "nwin");
    PyArrayObject* pr = nw ? as_array(pair_obj, NPY_INT32, 2, "pair_idx") : nullptr;
    PyObject* d = nullptr;
    PyObject* size_obj = nullptr;
    if (nw && pr) {
        const npy_intp B = PyArray_DIM(mask, 0), VL = PyArray_DIM(mask, 1), MB = PyArray_DIM(mask, 2);
        const npy_intp P = PyArray_DIM(pr, 0);
        const int64_t* nwin = (const int64_t*)PyArray_DATA(nw);
        const int32_t* pair = (const int32_t*)PyArray_DATA(pr);
        const uint8_t* m = (const uint8_t*)PyArray_DATA(mask);
        bool ok = PyArray_DIM(pr, 1) == 2 && PyArray_DIM(nw, 0) == B && MB == (P + 7) / 8;
        Py_ssize_t total = 0;
        for (npy_intp b = 0; ok && b < B; ++b) {
            if (nwin[b] < 0 || nwin[b] > VL) ok = false;
            total += (Py_ssize_t)nwin[b];
        }
        if (!ok || k0 < 0 || nkeys < k0 + total) {
            PyErr_SetString(PyExc_ValueError, "build_stdict: inconsistent shapes (mask / nwin / pair_idx / keys)");
        } else {
            if (into != Py_None) { d = into; Py_INCREF(d); } else d = PyDict_New();
            const bool ranged = ru0 >= 0 && ru1 >= ru0;
            // 'size' goes right after the first band of the first group; of a streamed pass: with the unit range that
            // holds the first band's last window (or the first range, if that band has no windows)
            const bool fresh = d && (ranged ? (B > 0 && ((ru0 < nwin[0] && nwin[0] <= ru1) || (nwin[0] == 0 && ru0 == 0)))
                                            : PyDict_Size(d) == 0);
            size_obj = PyLong_FromLong(nchans);
            std::vector<int32_t> dropped((size_t)P);
            PatternCache local;
            if (cache) {                        // another pair table than last time: the cached arrays do not apply
                uint64_t ph = 1469598103934665603ull;
                for (npy_intp i = 0; i < 2 * P; ++i) ph = (ph ^ (uint64_t)(uint32_t)pair[i]) * 1099511628211ull;
                if (cache->P != P || cache->pair_hash != ph) { cache->clear(); cache->P = P; cache->pair_hash = ph; }
            }
            auto& patterns = cache ? cache->patterns : local.patterns;       // mask bytes -> shared value array (owned)
            auto& patterns64 = cache ? cache->patterns64 : local.patterns64; // the same for up to 64 pairs: the mask IS the key
            std::string pat((size_t)MB, '\0');
            const uint8_t* last = nullptr;                            // run of equal masks: skip the lookup
            PyObject* last_arr = nullptr;
            Py_ssize_t k = k0;
            Py_ssize_t uflat = 0;                                     // flat unit index of (b, 0)
            for (npy_intp b = 0; b < B && d; uflat += (Py_ssize_t)nwin[b], ++b) {
                int64_t wlo = 0, whi = nwin[b];
                if (ranged) {
                    if (uflat + nwin[b] <= ru0 || uflat >= ru1) {      // band outside the range
                        k += (Py_ssize_t)nwin[b];
                        if (b == 0 && fresh && PyDict_SetItemString(d, "size", size_obj)) Py_CLEAR(d);
                        continue;
                    }
                    if (ru0 > uflat) wlo = ru0 - uflat;
                    if (ru1 < uflat + nwin[b]) whi = ru1 - uflat;
                    k += (Py_ssize_t)wlo;
                }
                for (int64_t w = wlo; w < whi && d; ++w, ++k) {
                    const uint8_t* mm = m + ((size_t)b * VL + (size_t)w) * MB;
                    PyObject* arr = nullptr;
                    if (last && memcmp(last, mm, (size_t)MB) == 0) {
                        arr = last_arr;
                    } else {
                        int c = 0;
                        for (npy_intp by = 0; by < MB; ++by) {
                            unsigned int z = (~(unsigned int)mm[by]) & 0xffu;          // zero-weight pairs of this byte
                            if (by == MB - 1 && (P & 7)) z &= (1u << (P & 7)) - 1u;     // bits beyond P are padding
                            pat[(size_t)by] = (char)z;
                            while (z) {
                                const int bit = __builtin_ctz(z);
                                z &= z - 1;
                                dropped[c++] = (int32_t)(8 * by + bit);
                            }
                        }
                        if (c) {
                            uint64_t key64 = 0;
                            if (MB <= 8) {
                                memcpy(&key64, pat.data(), (size_t)MB);
                                auto it64 = patterns64.find(key64);
                                if (it64 != patterns64.end()) arr = it64->second;
                            } else {
                                auto it = patterns.find(pat);
                                if (it != patterns.end()) arr = it->second;
                            }
                            if (!arr) {
                                npy_intp dims[1] = {2 * c};
                                arr = PyArray_SimpleNew(1, dims, NPY_INT32);
                                if (!arr) { Py_CLEAR(d); break; }
                                int32_t* a = (int32_t*)PyArray_DATA((PyArrayObject*)arr);
                                for (int i = 0; i < c; ++i) {
                                    a[i] = pair[2 * dropped[i]] + 1;
                                    a[c + i] = pair[2 * dropped[i] + 1] + 1;
                                }
                                PyArray_CLEARFLAGS((PyArrayObject*)arr, NPY_ARRAY_WRITEABLE);
                                if (MB <= 8) patterns64.emplace(key64, arr); else patterns.emplace(pat, arr);
                            }
                        }
                        last = mm;
                        last_arr = arr;
                    }
                    if (!arr) continue;                 // nothing dropped in this window: no entry
                    if (ktext) {
                        // the key object is made here, for the windows that need one only (ASCII: one memcpy)
                        const Py_ssize_t n = ((const uint8_t*)PyArray_DATA(klen))[k];
                        PyObject* ks = PyUnicode_New(n, 127);
                        if (!ks) { Py_CLEAR(d); break; }
                        memcpy(PyUnicode_DATA(ks), (const char*)PyArray_DATA(ktext) + (size_t)k * KEYW, (size_t)n);
                        const int bad = PyDict_SetItem(d, ks, arr);
                        Py_DECREF(ks);
                        if (bad) { Py_CLEAR(d); break; }
                    } else if (PyDict_SetItem(d, PyList_GET_ITEM(keys, k), arr)) { Py_CLEAR(d); break; }
                }
                k += (Py_ssize_t)(nwin[b] - whi);
                if (d && b == 0 && fresh && PyDict_SetItemString(d, "size", size_obj)) Py_CLEAR(d);
            }
            local.clear();                      // (a caller-held cache keeps its arrays)
            if (d && B == 0 && !ranged && fresh && PyDict_SetItemString(d, "size", size_obj)) Py_CLEAR(d);
        }
    }
    Py_XDECREF(size_obj);
    Py_DECREF(mask);
    Py_XDECREF(nw);
    Py_XDECREF(pr);
    Py_XDECREF(ktext);
    Py_XDECREF(klen);
    return d;
}

// An empty dict with room for n entries: the dictionary of a call grows to ~5*10^4 keys; without the hint it is
// re-hashed a dozen times on the way (every resize re-inserts all entries).
extern "C" PyObject* _PyDict_NewPresized(Py_ssize_t minused);
PyObject* new_dict(PyObject*, PyObject* arg) {
    const Py_ssize_t n = PyLong_AsSsize_t(arg);
    if (n == -1 && PyErr_Occurred()) return nullptr;
    return _PyDict_NewPresized(n > 0 ? n : 0);
}

PyMethodDef methods[] = {
    {"new_dict", new_dict, METH_O, "new_dict(n) -> empty dict presized for n entries"},
    {"new_pattern_cache", new_pattern_cache, METH_NOARGS, "new_pattern_cache() -> object for build_stdict(..., cache): value arrays shared across its calls"},
    {"float_repr", float_repr, METH_O, "repr(float) computed by this module (self-test hook)"},
    {"time_keys", time_keys, METH_VARARGS, "time_keys(t, nwin, prefixes) -> flat list of key strings"},
    {"time_key_text", time_key_text, METH_VARARGS, "time_key_text(t, nwin, prefixes, nthreads=1) -> (text (K, 40) uint8, length (K,) uint8)"},
    {"build_stdict", build_stdict, METH_VARARGS, "build_stdict(mask, nwin, pair_idx, nchans, keys) -> dict"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_nbls_host", "host-side helpers of narrow_band_least_squares_amd", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__nbls_host(void) {
    import_array();
    return PyModule_Create(&moddef);
}
