/* _rpfast: the per-cycle foreign call of ReactivePlanner.plan() as a CPython extension.
 *
 * The binding of the C ABI is ctypes (commonroad_rp_amd/_capi.py); for the one call a replanning cycle makes -- rp_plan with
 * RP_PLAN_PACKED (include/rp_amd.h; the reference's cycle: reactive_planner.py:570-665) -- ctypes costs about as much as the rest of
 * the cycle's Python: six struct fields written through descriptors, four slice assignments into the context's grid buffer, the
 * marshalling of nine arguments (profiles/probe_plan_split_r05.py: 5 us of a 45-us plan()).  This module does those steps in C:
 * it writes the cycle's fields into the caller's rp_params, copies the grids into the context's buffer (rp_fast_buffer) and calls
 * rp_plan through the function pointer the ctypes binding hands over -- the module is NOT linked against librp_amd.so, the library
 * stays loaded once, by _capi.load_library().  Everything else (the result struct, the output block, error texts) is read on the
 * Python side as before.  Without this module the binding takes its ctypes path: same calls, same results.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#include "rp_amd.h"

typedef int (*rp_plan_fn)(rp_ctx *, const rp_params *, const rp_cost *, const rp_grids *, int64_t, int64_t, uint32_t, rp_result *, double *);

typedef struct {
    PyObject_HEAD
    rp_plan_fn plan;
    rp_ctx *ctx;
    double *buf;       /* rp_fast_buffer: T | L | D as doubles, traj_len as int32 behind 2 * (nT + nL + nD) int32 words */
    size_t buf_bytes;
    rp_result *res;    /* the binding's result struct (struct_size set by its owner) */
    rp_grids dims;     /* sizes only: RP_PLAN_PACKED reads the arrays from the context's buffer */
} PackedObject;

static int packed_init(PackedObject *self, PyObject *args, PyObject *kw) {
    unsigned long long fn = 0, ctx = 0, buf = 0, nbytes = 0, res = 0;
    if (!PyArg_ParseTuple(args, "KKKKK", &fn, &ctx, &buf, &nbytes, &res)) return -1;
    if (!fn || !ctx || !buf || !res) { PyErr_SetString(PyExc_ValueError, "_rpfast.Packed: null address"); return -1; }
    self->plan = (rp_plan_fn)(uintptr_t)fn;
    self->ctx = (rp_ctx *)(uintptr_t)ctx;
    self->buf = (double *)(uintptr_t)buf;
    self->buf_bytes = (size_t)nbytes;
    self->res = (rp_result *)(uintptr_t)res;
    memset(&self->dims, 0, sizeof self->dims);
    self->dims.struct_size = (uint32_t)sizeof(rp_grids);
    return 0;
}

/* a C-contiguous 1-D buffer of `itemsize`-byte items -> pointer and length (items) */
static int view_of(PyObject *o, Py_buffer *v, Py_ssize_t itemsize, const char *what) {
    if (PyObject_GetBuffer(o, v, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return -1;
    const char *f = v->format ? v->format : "B";
    const int ok = v->itemsize == itemsize && v->ndim == 1 &&
                   (itemsize == 8 ? (f[0] == 'd' && f[1] == 0) : ((f[0] == 'i' || f[0] == 'l') && f[1] == 0));
    if (!ok) {
        PyBuffer_Release(v);
        PyErr_Format(PyExc_TypeError, "_rpfast.Packed.plan: %s must be a contiguous 1-D %s array", what, itemsize == 8 ? "float64" : "int32");
        return -1;
    }
    return 0;
}

/* the doubles of a grid argument -> dst (at most cap): a contiguous float64 array, or a Python set of floats in ITS iteration order
 * -- the order of the reference's loops over its sample sets, sampling.py:218-226 -- with `extra` (may be NULL) united in the way
 * `set.union({extra})` does it: a copy of the set, then the insertion.  Returns the count, -1 with an exception set. */
static Py_ssize_t grid_doubles(PyObject *o, PyObject *extra, double *dst, size_t cap, const char *what) {
    if (PyAnySet_Check(o)) {
        PyObject *u = o;
        Py_INCREF(u);
        if (extra) {
            /* always the copy, also when `extra` is in the set already: a copy's table is sized for the final count, the original
               grew, and the two need not iterate in the same order (17 floats: they do not) -- the reference iterates the copy */
            Py_DECREF(u);
            u = PySet_New(o);
            if (!u) return -1;
            if (PySet_Add(u, extra) != 0) { Py_DECREF(u); return -1; }
        }
        PyObject *it = PyObject_GetIter(u);
        Py_DECREF(u);   /* (the iterator keeps the set) */
        if (!it) return -1;
        Py_ssize_t k = 0;
        PyObject *item;
        while ((item = PyIter_Next(it)) != NULL) {
            const double v = PyFloat_AsDouble(item);
            Py_DECREF(item);
            if (v == -1.0 && PyErr_Occurred()) { Py_DECREF(it); return -1; }
            if ((size_t)k >= cap) { Py_DECREF(it); PyErr_SetString(PyExc_ValueError, "_rpfast.Packed.plan: grids larger than the context's buffer"); return -1; }
            dst[k++] = v;
        }
        Py_DECREF(it);
        return PyErr_Occurred() ? -1 : k;
    }
    Py_buffer v;
    if (view_of(o, &v, 8, what) != 0) return -1;
    const Py_ssize_t k = v.len / 8;
    if ((size_t)k > cap) { PyBuffer_Release(&v); PyErr_SetString(PyExc_ValueError, "_rpfast.Packed.plan: grids larger than the context's buffer"); return -1; }
    memcpy(dst, v.buf, (size_t)k * 8);
    PyBuffer_Release(&v);
    return k;
}

static int three(PyObject *seq, double *dst, const char *what) {
    PyObject *fast = PySequence_Fast(seq, what);
    if (!fast) return -1;
    if (PySequence_Fast_GET_SIZE(fast) != 3) { Py_DECREF(fast); PyErr_Format(PyExc_ValueError, "%s: need three values", what); return -1; }
    for (int k = 0; k < 3; ++k) {
        dst[k] = PyFloat_AsDouble(PySequence_Fast_GET_ITEM(fast, k));
        if (dst[k] == -1.0 && PyErr_Occurred()) { Py_DECREF(fast); return -1; }
    }
    Py_DECREF(fast);
    return 0;
}

/* plan(params, cost, T, traj_len, L, D, out, time_step0, low_vel_mode, flags, x0_lon, x0_lat, orientation) -> rc
 * params / cost: the caller's rp_params (writable) / rp_cost as buffer objects (ctypes structures); `out`: writable float64 buffer
 * of (RP_N_ARRAYS + 13) * (N + 1) doubles.  L, D: float64 arrays or Python sets of floats (grid_doubles).  The rp_result written is
 * the one given to the constructor. */
static PyObject *packed_plan(PackedObject *self, PyObject *const *args, Py_ssize_t nargs) {
    if (nargs != 13) { PyErr_SetString(PyExc_TypeError, "_rpfast.Packed.plan takes 13 arguments"); return NULL; }
    /* the caller's structs (ctypes structures: the buffer protocol gives their addresses; they stay the caller's) */
    Py_buffer vp, vc;
    if (PyObject_GetBuffer(args[0], &vp, PyBUF_WRITABLE) != 0) return NULL;
    if (PyObject_GetBuffer(args[1], &vc, PyBUF_SIMPLE) != 0) { PyBuffer_Release(&vp); return NULL; }
    rp_params *p = (rp_params *)vp.buf;
    const rp_cost *cost = (const rp_cost *)vc.buf;
    const int sized = (size_t)vp.len >= sizeof(rp_params) && (size_t)vc.len >= sizeof(rp_cost);
    PyBuffer_Release(&vp); PyBuffer_Release(&vc);   /* (the objects outlive this call: arguments of it) */
    if (!sized) { PyErr_SetString(PyExc_ValueError, "_rpfast.Packed.plan: params / cost are not rp_params / rp_cost of this ABI"); return NULL; }
    const long t0 = PyLong_AsLong(args[7]);
    if (t0 == -1 && PyErr_Occurred()) return NULL;
    const int low = PyObject_IsTrue(args[8]);
    if (low < 0) return NULL;
    const unsigned long flags = PyLong_AsUnsignedLong(args[9]);
    if (flags == (unsigned long)-1 && PyErr_Occurred()) return NULL;
    double lon[3], lat[3];
    if (three(args[10], lon, "x0_lon") != 0 || three(args[11], lat, "x0_lat") != 0) return NULL;
    const double orientation = PyFloat_AsDouble(args[12]);
    if (orientation == -1.0 && PyErr_Occurred()) return NULL;

    Py_buffer vT, vtl, vout;
    if (view_of(args[2], &vT, 8, "T") != 0) return NULL;
    if (view_of(args[3], &vtl, 4, "traj_len") != 0) { PyBuffer_Release(&vT); return NULL; }
    PyObject *ret = NULL;
    if (PyObject_GetBuffer(args[6], &vout, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) goto release2;
    {
        const size_t nT = (size_t)(vT.len / 8);
        const size_t n = (size_t)p->N + 1;
        /* the buffer holds [T | L | D] doubles and nT int32 words behind them (rp_amd.h: RP_PLAN_PACKED) */
        const size_t cap_words = self->buf_bytes / 8;
        if ((size_t)(vtl.len / 4) != nT) { PyErr_SetString(PyExc_ValueError, "_rpfast.Packed.plan: traj_len and T differ in length"); goto release3; }
        if (nT + (nT + 1) / 2 > cap_words) { PyErr_SetString(PyExc_ValueError, "_rpfast.Packed.plan: grids larger than the context's buffer"); goto release3; }
        if ((size_t)vout.len < (size_t)(RP_N_ARRAYS + 13) * n * sizeof(double)) { PyErr_SetString(PyExc_ValueError, "_rpfast.Packed.plan: output block too small"); goto release3; }
        /* the grids -> the context's buffer, in the layout RP_PLAN_PACKED reads (rp_amd.h: rp_fast_buffer).  L / D: arrays, or the
           sample SETS themselves (D: united with the current lateral offset, sampling.py:226) */
        double *b = self->buf;
        const size_t room = cap_words - nT - (nT + 1) / 2;
        memcpy(b, vT.buf, nT * 8);
        const Py_ssize_t nLs = grid_doubles(args[4], NULL, b + nT, room, "L");
        if (nLs < 0) goto release3;
        const size_t nL = (size_t)nLs;
        PyObject *d0 = NULL;
        if (PyAnySet_Check(args[5])) {
            d0 = PySequence_GetItem(args[11], 0);
            if (!d0) goto release3;
        }
        const Py_ssize_t nDs = grid_doubles(args[5], d0, b + nT + nL, room - nL, "D");
        Py_XDECREF(d0);
        if (nDs < 0) goto release3;
        const size_t nD = (size_t)nDs, nd = nT + nL + nD;
        memcpy((int32_t *)b + 2 * nd, vtl.buf, nT * 4);
        /* the cycle's fields of rp_params (reactive_planner.py:586-594: time step, low-velocity flag, curvilinear state, orientation) */
        p->time_step0 = (int32_t)t0;
        p->low_vel_mode = low;
        p->flags = (uint32_t)flags;
        memcpy(p->x0_lon, lon, sizeof lon);
        memcpy(p->x0_lat, lat, sizeof lat);
        p->x0_orientation = orientation;
        self->dims.nT = (int32_t)nT; self->dims.nL = (int32_t)nL; self->dims.nD = (int32_t)nD;
        int rc = 0;
        if (nT * nL * nD != 0) {   /* (an empty bundle is the caller's to skip; the entry would refuse it) */
            Py_BEGIN_ALLOW_THREADS
            rc = self->plan(self->ctx, p, cost, &self->dims, 0, -1, RP_PLAN_PACKED, self->res, (double *)vout.buf);
            Py_END_ALLOW_THREADS
        } else {
            rc = RP_EINVAL;
        }
        ret = PyLong_FromLong(rc);
    }
release3:
    PyBuffer_Release(&vout);
release2:
    PyBuffer_Release(&vT); PyBuffer_Release(&vtl);
    return ret;
}

static PyMethodDef packed_methods[] = {
    {"plan", (PyCFunction)(void (*)(void))packed_plan, METH_FASTCALL,
     "plan(params, cost, T, traj_len, L, D, out, time_step0, low_vel_mode, flags, x0_lon, x0_lat, orientation) -> rc of rp_plan(RP_PLAN_PACKED)"},
    {NULL, NULL, 0, NULL}};

static PyTypeObject PackedType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "_rpfast.Packed",
    .tp_basicsize = sizeof(PackedObject),
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_doc = "Packed(rp_plan address, context handle, fast buffer address, its bytes, rp_result address): one sampling level per call",
    .tp_methods = packed_methods,
    .tp_init = (initproc)packed_init,
    .tp_new = PyType_GenericNew,
};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_rpfast", "per-cycle foreign call of ReactivePlanner.plan() without ctypes (see rp_pyfast.c)", -1, NULL};

PyMODINIT_FUNC PyInit__rpfast(void) {
    if (PyType_Ready(&PackedType) < 0) return NULL;
    PyObject *m = PyModule_Create(&moduledef);
    if (!m) return NULL;
    Py_INCREF(&PackedType);
    if (PyModule_AddObject(m, "Packed", (PyObject *)&PackedType) < 0) { Py_DECREF(&PackedType); Py_DECREF(m); return NULL; }
    PyModule_AddIntConstant(m, "ABI_VERSION", RP_ABI_VERSION);
    PyModule_AddIntConstant(m, "SIZEOF_PARAMS", (long)sizeof(rp_params));
    PyModule_AddIntConstant(m, "SIZEOF_RESULT", (long)sizeof(rp_result));
    return m;
}
