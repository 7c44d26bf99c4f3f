// Semiring operators for the device kernels.
//
// A kernel is a template over a semiring class S that provides
//   S::x_t    type of the values read from A and B (mult->xtype == mult->ytype)
//   S::z_t    type of the products and of C (monoid type)
//   S::acc_t  storage type of an accumulator slot (>= 32 bit, so that LDS and
//             global atomics exist for it; narrowed to z_t on output)
//   mult(a, b, i, k, j)   t = A(i,k) (x) B(k,j)
//   add(x, y)             monoid
//   atomic(acc_t *p, t)   *p = add(*p, t), atomically
//   identity()
// Two families: Sct<ADD,MULT,XT,ZT> fixes the operators at compile time (the
// hot semirings); Srt<XT,ZT> carries the opcodes at run time and covers the
// rest of the built-in set of Source/Factories/GB_AxB_factory.c:42-372 with a
// wave-uniform switch.
//
// Arithmetic follows the reference: integer +,* wrap (gcc -fwrapv,
// cmake_modules/GraphBLAS_compiler_options.cmake:23-29); integer division is
// GB_idiv_* (Source/GB_math.h:27-260); fp MIN/MAX are fmin/fmax, i.e. they
// omit NaN (Source/FactoryKernels/GB_AxB__min_plus_fp64.c:16-19); the boolean
// renames of Source/GB_boolean_rename.c:45-97 are applied on the host before a
// semiring reaches a kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/graphblas_amd.h"

namespace gba {

// ------------------------------------------------------------ type traits --
template <class T> struct acc_of { using type = T; };
template <> struct acc_of<bool>     { using type = uint32_t; };
template <> struct acc_of<int8_t>   { using type = int32_t; };
template <> struct acc_of<int16_t>  { using type = int32_t; };
template <> struct acc_of<uint8_t>  { using type = uint32_t; };
template <> struct acc_of<uint16_t> { using type = uint32_t; };

template <class T> struct type_code;
template <> struct type_code<bool>     { static constexpr int v = GBA_BOOL; };
template <> struct type_code<int8_t>   { static constexpr int v = GBA_INT8; };
template <> struct type_code<uint8_t>  { static constexpr int v = GBA_UINT8; };
template <> struct type_code<int16_t>  { static constexpr int v = GBA_INT16; };
template <> struct type_code<uint16_t> { static constexpr int v = GBA_UINT16; };
template <> struct type_code<int32_t>  { static constexpr int v = GBA_INT32; };
template <> struct type_code<uint32_t> { static constexpr int v = GBA_UINT32; };
template <> struct type_code<int64_t>  { static constexpr int v = GBA_INT64; };
template <> struct type_code<uint64_t> { static constexpr int v = GBA_UINT64; };
template <> struct type_code<float>    { static constexpr int v = GBA_FP32; };
template <> struct type_code<double>   { static constexpr int v = GBA_FP64; };

__host__ __device__ inline size_t type_size(int code) {
    switch (code) {
        case GBA_BOOL: case GBA_INT8: case GBA_UINT8: return 1;
        case GBA_INT16: case GBA_UINT16: return 2;
        case GBA_INT32: case GBA_UINT32: case GBA_FP32: return 4;
        case GBA_INT64: case GBA_UINT64: case GBA_FP64: case GBA_FC32: return 8;
        case GBA_FC64: return 16;
        default: return 0;
    }
}

template <class T> __host__ __device__ inline T tmax_value() {
    if constexpr (std::is_same<T, float>::value) return __builtin_huge_valf();
    else if constexpr (std::is_same<T, double>::value) return __builtin_huge_val();
    else if constexpr (std::is_same<T, bool>::value) return true;
    else if constexpr (std::is_signed<T>::value) return (T)((((uint64_t)1) << (8 * sizeof(T) - 1)) - 1);
    else return (T)~(T)0;
}
template <class T> __host__ __device__ inline T tmin_value() {
    if constexpr (std::is_same<T, float>::value) return -__builtin_huge_valf();
    else if constexpr (std::is_same<T, double>::value) return -__builtin_huge_val();
    else if constexpr (std::is_same<T, bool>::value) return false;
    else if constexpr (std::is_signed<T>::value) return (T)(-(int64_t)((((uint64_t)1) << (8 * sizeof(T) - 1)) - 1) - 1);
    else return (T)0;
}

// ------------------------------------------------------- scalar operators --
template <class T> __host__ __device__ inline T idiv(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) {
        return x / y;
    } else if constexpr (std::is_same<T, bool>::value) {
        return x;                                   // bool DIV is FIRST after renaming
    } else if constexpr (std::is_signed<T>::value) {
        if (y == (T)-1) return (T)(0 - (typename std::make_unsigned<T>::type)x);
        if (y == 0) return (x == 0) ? (T)0 : ((x < 0) ? tmin_value<T>() : tmax_value<T>());
        return (T)(x / y);
    } else {
        if (y == 0) return (x == 0) ? (T)0 : tmax_value<T>();
        return (T)(x / y);
    }
}

template <class T> __host__ __device__ inline T op_min(T x, T y) {
    if constexpr (std::is_same<T, float>::value) return fminf(x, y);
    else if constexpr (std::is_same<T, double>::value) return fmin(x, y);
    else return (x < y) ? x : y;
}
template <class T> __host__ __device__ inline T op_max(T x, T y) {
    if constexpr (std::is_same<T, float>::value) return fmaxf(x, y);
    else if constexpr (std::is_same<T, double>::value) return fmax(x, y);
    else return (x > y) ? x : y;
}
// wrap-around integer arithmetic without signed-overflow UB
template <class T> __host__ __device__ inline T op_plus(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return x + y;
    else if constexpr (std::is_same<T, bool>::value) return x || y;
    else { using U = typename std::make_unsigned<T>::type; return (T)((U)x + (U)y); }
}
template <class T> __host__ __device__ inline T op_minus(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return x - y;
    else if constexpr (std::is_same<T, bool>::value) return x != y;
    else { using U = typename std::make_unsigned<T>::type; return (T)((U)x - (U)y); }
}
template <class T> __host__ __device__ inline T op_times(T x, T y) {
    if constexpr (std::is_floating_point<T>::value) return x * y;
    else if constexpr (std::is_same<T, bool>::value) return x && y;
    else if constexpr (sizeof(T) < 4) return (T)((uint32_t)x * (uint32_t)y);
    else { using U = typename std::make_unsigned<T>::type; return (T)((U)x * (U)y); }
}

// z = mult(x,y) for multipliers whose result has the type of their inputs
template <class T> __host__ __device__ inline T mult_tt(int op, T a, T b) {
    switch (op) {
        case GBA_FIRST:  return a;
        case GBA_SECOND: return b;
        case GBA_ANY:    return a;
        case GBA_PAIR:   return (T)1;
        case GBA_MIN:    return op_min<T>(a, b);
        case GBA_MAX:    return op_max<T>(a, b);
        case GBA_PLUS:   return op_plus<T>(a, b);
        case GBA_MINUS:  return op_minus<T>(a, b);
        case GBA_RMINUS: return op_minus<T>(b, a);
        case GBA_TIMES:  return op_times<T>(a, b);
        case GBA_DIV:    return idiv<T>(a, b);
        case GBA_RDIV:   return idiv<T>(b, a);
        case GBA_ISEQ:   return (T)(a == b);
        case GBA_ISNE:   return (T)(a != b);
        case GBA_ISGT:   return (T)(a > b);
        case GBA_ISLT:   return (T)(a < b);
        case GBA_ISGE:   return (T)(a >= b);
        case GBA_ISLE:   return (T)(a <= b);
        case GBA_LOR:    return (T)((a != (T)0) || (b != (T)0));
        case GBA_LAND:   return (T)((a != (T)0) && (b != (T)0));
        case GBA_LXOR:   return (T)((a != (T)0) != (b != (T)0));
        case GBA_EQ:     return (T)(a == b);          // bool only (LXNOR)
        case GBA_NE:     return (T)(a != b);
        case GBA_GT:     return (T)(a > b);
        case GBA_LT:     return (T)(a < b);
        case GBA_GE:     return (T)(a >= b);
        case GBA_LE:     return (T)(a <= b);
        default: break;
    }
    if constexpr (std::is_integral<T>::value && !std::is_same<T, bool>::value) {
        switch (op) {
            case GBA_BOR:   return (T)(a | b);
            case GBA_BAND:  return (T)(a & b);
            case GBA_BXOR:  return (T)(a ^ b);
            case GBA_BXNOR: return (T)~(a ^ b);
            default: break;
        }
    }
    return (T)0;
}

// comparators: T x T -> bool
template <class T> __host__ __device__ inline bool mult_cmp(int op, T a, T b) {
    switch (op) {
        case GBA_EQ: return a == b;
        case GBA_NE: return a != b;
        case GBA_GT: return a > b;
        case GBA_LT: return a < b;
        case GBA_GE: return a >= b;
        case GBA_LE: return a <= b;
        default: return false;
    }
}

// positional multipliers: C(i,j) += A(i,k) (x) B(k,j)
template <class Z> __host__ __device__ inline Z mult_pos(int op, int64_t i, int64_t k, int64_t j) {
    switch (op) {
        case GBA_FIRSTI:   return (Z)i;
        case GBA_FIRSTI1:  return (Z)(i + 1);
        case GBA_FIRSTJ:   case GBA_SECONDI:  return (Z)k;
        case GBA_FIRSTJ1:  case GBA_SECONDI1: return (Z)(k + 1);
        case GBA_SECONDJ:  return (Z)j;
        case GBA_SECONDJ1: return (Z)(j + 1);
        default: return (Z)0;
    }
}
__host__ __device__ inline bool op_is_positional(int op) { return op >= GBA_FIRSTI && op <= GBA_SECONDJ1; }

// monoid z = add(x,y)
template <class T> __host__ __device__ inline T add_tt(int op, T x, T y) {
    switch (op) {
        case GBA_ANY:   return x;
        case GBA_MIN:   return op_min<T>(x, y);
        case GBA_MAX:   return op_max<T>(x, y);
        case GBA_PLUS:  return op_plus<T>(x, y);
        case GBA_TIMES: return op_times<T>(x, y);
        case GBA_LOR:   return (T)((x != (T)0) || (y != (T)0));
        case GBA_LAND:  return (T)((x != (T)0) && (y != (T)0));
        case GBA_LXOR:  return (T)((x != (T)0) != (y != (T)0));
        case GBA_EQ:    return (T)(x == y);
        default: break;
    }
    if constexpr (std::is_integral<T>::value && !std::is_same<T, bool>::value) {
        switch (op) {
            case GBA_BOR:   return (T)(x | y);
            case GBA_BAND:  return (T)(x & y);
            case GBA_BXOR:  return (T)(x ^ y);
            case GBA_BXNOR: return (T)~(x ^ y);
            default: break;
        }
    }
    return x;
}

template <class T> __host__ __device__ inline T identity_tt(int op) {
    switch (op) {
        case GBA_MIN:   return tmax_value<T>();
        case GBA_MAX:   return tmin_value<T>();
        case GBA_TIMES: return (T)1;
        case GBA_LAND:  return (T)1;
        case GBA_EQ:    return (T)1;
        default: break;
    }
    if constexpr (std::is_integral<T>::value && !std::is_same<T, bool>::value) {
        if (op == GBA_BAND || op == GBA_BXNOR) return (T)~(T)0;
    }
    return (T)0;                                    // PLUS, ANY, LOR, LXOR, BOR, BXOR
}

// ---------------------------------------------------------------- atomics --
// *p = add(*p, t) on an accumulator slot in LDS or global memory.
template <class A> __device__ inline A atomic_cas(A *p, A expected, A desired) {
    if constexpr (sizeof(A) == 4) {
        unsigned int r = atomicCAS(reinterpret_cast<unsigned int *>(p),
                                   *reinterpret_cast<unsigned int *>(&expected),
                                   *reinterpret_cast<unsigned int *>(&desired));
        return *reinterpret_cast<A *>(&r);
    } else {
        unsigned long long r = atomicCAS(reinterpret_cast<unsigned long long *>(p),
                                         *reinterpret_cast<unsigned long long *>(&expected),
                                         *reinterpret_cast<unsigned long long *>(&desired));
        return *reinterpret_cast<A *>(&r);
    }
}

template <class A> __device__ inline bool bits_equal(A a, A b) {
    if constexpr (sizeof(A) == 4) return *reinterpret_cast<uint32_t *>(&a) == *reinterpret_cast<uint32_t *>(&b);
    else return *reinterpret_cast<uint64_t *>(&a) == *reinterpret_cast<uint64_t *>(&b);
}

template <class Z, class A, class F> __device__ inline void atomic_cas_loop(A *p, F f) {
    A old = *reinterpret_cast<volatile A *>(p);
    while (true) {
        const A nw = (A)f((Z)old);
        if (bits_equal<A>(nw, old)) return;
        const A seen = atomic_cas<A>(p, old, nw);
        if (bits_equal<A>(seen, old)) return;
        old = seen;
    }
}

template <class Z> __device__ inline void atomic_monoid(int op, typename acc_of<Z>::type *p, Z t) {
    using A = typename acc_of<Z>::type;
    if constexpr (std::is_floating_point<Z>::value) {
        switch (op) {
            case GBA_PLUS: atomicAdd(p, t); return;
            case GBA_ANY:  *reinterpret_cast<volatile A *>(p) = t; return;
            case GBA_MIN:  if (t == t) atomic_cas_loop<Z, A>(p, [t](Z z) { return op_min<Z>(z, t); }); return;
            case GBA_MAX:  if (t == t) atomic_cas_loop<Z, A>(p, [t](Z z) { return op_max<Z>(z, t); }); return;
            case GBA_TIMES: atomic_cas_loop<Z, A>(p, [t](Z z) { return z * t; }); return;
            default: return;
        }
    } else if constexpr (sizeof(A) == 4) {
        using S = int32_t; using U = uint32_t;
        A a = (A)t;
        switch (op) {
            case GBA_PLUS:  atomicAdd(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_ANY:   *reinterpret_cast<volatile A *>(p) = a; return;
            case GBA_MIN:   if constexpr (std::is_signed<A>::value) atomicMin(reinterpret_cast<S *>(p), (S)a);
                            else atomicMin(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_MAX:   if constexpr (std::is_signed<A>::value) atomicMax(reinterpret_cast<S *>(p), (S)a);
                            else atomicMax(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_TIMES: atomic_cas_loop<A, A>(p, [a](A z) { return (A)((U)z * (U)a); }); return;
            case GBA_LOR:   if (a) atomicOr(reinterpret_cast<U *>(p), 1u); return;
            case GBA_LAND:  if (!a) atomicAnd(reinterpret_cast<U *>(p), 0u); return;
            case GBA_LXOR:  if (a) atomicXor(reinterpret_cast<U *>(p), 1u); return;
            case GBA_EQ:    if (!a) atomicXor(reinterpret_cast<U *>(p), 1u); return;   // xnor(z,t) = z ^ !t
            case GBA_BOR:   atomicOr(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_BAND:  atomicAnd(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_BXOR:  atomicXor(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_BXNOR: atomicXor(reinterpret_cast<U *>(p), ~(U)a); return;
            default: return;
        }
    } else {
        using S = long long; using U = unsigned long long;
        A a = (A)t;
        switch (op) {
            case GBA_PLUS:  atomicAdd(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_ANY:   *reinterpret_cast<volatile A *>(p) = a; return;
            case GBA_MIN:   if constexpr (std::is_signed<A>::value) atomicMin(reinterpret_cast<S *>(p), (S)a);
                            else atomicMin(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_MAX:   if constexpr (std::is_signed<A>::value) atomicMax(reinterpret_cast<S *>(p), (S)a);
                            else atomicMax(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_TIMES: atomic_cas_loop<A, A>(p, [a](A z) { return (A)((U)z * (U)a); }); return;
            case GBA_BOR:   atomicOr(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_BAND:  atomicAnd(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_BXOR:  atomicXor(reinterpret_cast<U *>(p), (U)a); return;
            case GBA_BXNOR: atomicXor(reinterpret_cast<U *>(p), ~(U)a); return;
            default: return;
        }
    }
}

// ------------------------------------------------------- semiring classes --
enum { KIND_TT = 0, KIND_CMP = 1, KIND_POS = 2 };

template <class XT, class ZT> struct mult_kind {
    static constexpr int v = std::is_same<XT, ZT>::value ? KIND_TT : KIND_CMP;
};

// run-time operators.  KIND: TT z_t == x_t; CMP z_t == bool, x_t any; POS x_t unused.
template <class XT, class ZT, int KIND>
struct Srt {
    using x_t = XT; using z_t = ZT; using acc_t = typename acc_of<ZT>::type;
    static constexpr bool positional = (KIND == KIND_POS);   // the multiplier reads i, k or j
    int add_op, mult_op;
    bool a_pattern, b_pattern;
    __host__ __device__ bool a_is_pattern() const { return a_pattern; }
    __host__ __device__ bool b_is_pattern() const { return b_pattern; }
    __device__ ZT mult(XT a, XT b, int64_t i, int64_t k, int64_t j) const {
        if constexpr (KIND == KIND_POS) return mult_pos<ZT>(mult_op, i, k, j);
        else if constexpr (KIND == KIND_CMP) return mult_cmp<XT>(mult_op, a, b);
        else return mult_tt<ZT>(mult_op, a, b);
    }
    __device__ ZT add(ZT x, ZT y) const { return add_tt<ZT>(add_op, x, y); }
    __host__ __device__ ZT identity() const { return identity_tt<ZT>(add_op); }
    __device__ void atomic(acc_t *p, ZT t) const { atomic_monoid<ZT>(add_op, p, t); }
    __host__ __device__ int add_code() const { return add_op; }
    __host__ __device__ bool add_is_any() const { return add_op == GBA_ANY; }
};

// compile-time operators
template <int ADD, int MULT, class XT, class ZT>
struct Sct {
    using x_t = XT; using z_t = ZT; using acc_t = typename acc_of<ZT>::type;
    static constexpr bool POS = (MULT >= GBA_FIRSTI && MULT <= GBA_SECONDJ1);
    static constexpr bool positional = POS;
    __host__ __device__ bool a_is_pattern() const { return POS || MULT == GBA_SECOND || MULT == GBA_PAIR; }
    __host__ __device__ bool b_is_pattern() const { return POS || MULT == GBA_FIRST || MULT == GBA_PAIR; }
    __device__ ZT mult(XT a, XT b, int64_t i, int64_t k, int64_t j) const {
        if constexpr (POS) return mult_pos<ZT>(MULT, i, k, j);
        else if constexpr (!std::is_same<XT, ZT>::value) return mult_cmp<XT>(MULT, a, b);
        else return mult_tt<ZT>(MULT, a, b);
    }
    __device__ ZT add(ZT x, ZT y) const { return add_tt<ZT>(ADD, x, y); }
    __host__ __device__ ZT identity() const { return identity_tt<ZT>(ADD); }
    __device__ void atomic(acc_t *p, ZT t) const { atomic_monoid<ZT>(ADD, p, t); }
    __host__ __device__ int add_code() const { return ADD; }
    __host__ __device__ bool add_is_any() const { return ADD == GBA_ANY; }
};

// load a value, honouring iso (one shared value) and pattern-only operands
template <class T> __device__ inline T load_x(const T *x, int64_t p, bool iso, bool pattern) {
    if (pattern) return T();
    return x[iso ? 0 : p];
}

// combine the partial results of the `width` lanes of a group (width = 2^q <= 64);
// found tells whether a lane holds a value at all
template <class S>
__device__ inline void group_reduce(const S &sr, typename S::z_t &v, int &found, int width) {
    using Z = typename S::z_t;
    for (int d = width >> 1; d > 0; d >>= 1) {
        Z ov;
        const int of = __shfl_down(found, d, width);
        if constexpr (sizeof(Z) == 8) {
            long long raw = __shfl_down(*reinterpret_cast<long long *>(&v), d, width);
            ov = *reinterpret_cast<Z *>(&raw);
        } else if constexpr (std::is_same<Z, float>::value) {
            ov = __shfl_down(v, d, width);
        } else {
            ov = (Z)__shfl_down((int)v, d, width);
        }
        if (of) { v = found ? sr.add(v, ov) : ov; found = 1; }
    }
}

}  // namespace gba
