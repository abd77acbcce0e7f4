// _tensor_table: the geometry of a LIST of torch tensors in one pass (host plumbing of the Python layer, NOT part of the C ABI:
// include/dctfp.h takes plain pointers and sizes, and this is where a flush of thousands of proteins gets them from).
//
// A database flush hands over 2 x 2 048 embedding matrices and 2 048 contact maps as torch tensors.  Reading data_ptr / size /
// stride / dtype / device of each through Python costs ~ 0.12 us per attribute -- 6 ms per flush of 2 048 proteins, more than
// the GPU work of the flush (profiles/r05/flush_timeline_before.txt: "embedding tables" 6.3 ms, "maps" 2.0 ms).  Here the
// list is walked once in C++: ~ 20 ns per tensor.
//
//   fill(seq, ptrs_addr, meta_addr, n) -> number of entries that are tensors
//     seq       : list / tuple of objects
//     ptrs_addr : address of uint64[n]      -> data_ptr() (0 for a non-tensor)
//     meta_addr : address of int64[n][6]    -> {dim, size(0), size(1), stride(0), stride(1), code}; sizes / strides beyond dim are 0;
//                 code = scalar type (c10::ScalarType as int) | device type << 8 | (device index + 1) << 16; dim = -1: not a tensor
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <torch/csrc/autograd/python_variable.h>

#include <cstdint>
#include <exception>

namespace {

PyObject* fill(PyObject*, PyObject* args) {
    PyObject* seq_in = nullptr;
    unsigned long long ptrs_addr = 0, meta_addr = 0;
    Py_ssize_t n = 0;
    if (!PyArg_ParseTuple(args, "OKKn", &seq_in, &ptrs_addr, &meta_addr, &n)) return nullptr;
    PyObject* seq = PySequence_Fast(seq_in, "expected a list or tuple of tensors");
    if (!seq) return nullptr;
    if (PySequence_Fast_GET_SIZE(seq) != n) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_ValueError, "length of the sequence differs from the arrays'");
        return nullptr;
    }
    uint64_t* ptrs = reinterpret_cast<uint64_t*>(static_cast<uintptr_t>(ptrs_addr));
    int64_t* meta = reinterpret_cast<int64_t*>(static_cast<uintptr_t>(meta_addr));
    PyObject** items = PySequence_Fast_ITEMS(seq);
    Py_ssize_t n_tensors = 0;
    try {   // (a sparse or otherwise exotic tensor makes ATen throw from stride() / data_ptr(): a Python error, not std::terminate)
    for (Py_ssize_t i = 0; i < n; ++i) {
        int64_t* m = meta + 6 * i;
        if (!THPVariable_Check(items[i])) {
            ptrs[i] = 0;
            m[0] = -1;
            m[1] = m[2] = m[3] = m[4] = m[5] = 0;
            continue;
        }
        const at::Tensor& t = THPVariable_Unpack(items[i]);
        const int64_t dim = t.dim();
        ptrs[i] = t.defined() && t.has_storage() ? reinterpret_cast<uint64_t>(t.data_ptr()) : 0;
        m[0] = dim;
        m[1] = dim >= 1 ? t.size(0) : 0;
        m[2] = dim >= 2 ? t.size(1) : 0;
        m[3] = dim >= 1 ? t.stride(0) : 0;
        m[4] = dim >= 2 ? t.stride(1) : 0;
        const c10::Device dev = t.device();
        m[5] = (int64_t)t.scalar_type() | ((int64_t)dev.type() << 8) | ((int64_t)(dev.has_index() ? dev.index() + 1 : 0) << 16);
        ++n_tensors;
    }
    } catch (const std::exception& e) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
    Py_DECREF(seq);
    return PyLong_FromSsize_t(n_tensors);
}

PyObject* scalar_code(PyObject*, PyObject* arg) {   // the code of one tensor (what `fill` writes into meta[5]): for the Python side's constants
    if (!THPVariable_Check(arg)) {
        PyErr_SetString(PyExc_TypeError, "expected a tensor");
        return nullptr;
    }
    const at::Tensor& t = THPVariable_Unpack(arg);
    const c10::Device dev = t.device();
    return PyLong_FromLongLong((long long)t.scalar_type() | ((long long)dev.type() << 8) | ((long long)(dev.has_index() ? dev.index() + 1 : 0) << 16));
}

PyMethodDef methods[] = {{"fill", fill, METH_VARARGS, "geometry of a list of tensors into two numpy arrays (by address)"},
                         {"scalar_code", scalar_code, METH_O, "dtype / device code of one tensor"},
                         {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_tensor_table", "bulk tensor geometry (host plumbing)", -1, methods, nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__tensor_table(void) { return PyModule_Create(&module); }
