// TEST INFRASTRUCTURE ONLY (oracle). Stand-in for DirectX-Headers' <wsl/winadapter.h>, which is an
// un-vendored dependency of the reference (CMakeLists.txt:383-389) and is absent from this image.
// Provides just the Win32 vocabulary the reference's headers use on non-Windows builds
// (DirectXTex.h:33-37, DirectXTexP.h:139-143). Written from scratch for this repo.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cwchar>

typedef int32_t HRESULT;
typedef uint32_t UINT;
typedef uint32_t DWORD;
typedef int32_t BOOL;
typedef int32_t LONG;
typedef uint8_t BYTE;
typedef uint16_t WORD;
typedef void* HANDLE;
typedef wchar_t WCHAR;
typedef size_t SIZE_T;
typedef uint64_t UINT64;
typedef void* LPVOID;

struct GUID { uint32_t Data1; uint16_t Data2; uint16_t Data3; uint8_t Data4[8]; };
typedef const GUID& REFGUID;
inline bool operator==(const GUID& a, const GUID& b) { return std::memcmp(&a, &b, sizeof(GUID)) == 0; }
inline bool operator!=(const GUID& a, const GUID& b) { return !(a == b); }

#define S_OK            static_cast<HRESULT>(0)
#define S_FALSE         static_cast<HRESULT>(1)
#define E_FAIL          static_cast<HRESULT>(0x80004005)
#define E_INVALIDARG    static_cast<HRESULT>(0x80070057)
#define E_OUTOFMEMORY   static_cast<HRESULT>(0x8007000E)
#define E_POINTER       static_cast<HRESULT>(0x80004003)
#define E_ABORT         static_cast<HRESULT>(0x80004004)
#define E_NOTIMPL       static_cast<HRESULT>(0x80004001)
#define E_UNEXPECTED    static_cast<HRESULT>(0x8000FFFF)
#define E_BOUNDS        static_cast<HRESULT>(0x8000000B)
#define E_NOINTERFACE   static_cast<HRESULT>(0x80004002)
#define FAILED(hr)      (static_cast<HRESULT>(hr) < 0)
#define SUCCEEDED(hr)   (static_cast<HRESULT>(hr) >= 0)

#define __cdecl
#define UNREFERENCED_PARAMETER(x) (void)(x)
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define UINT32_MAX_ UINT32_MAX

#define DEFINE_ENUM_FLAG_OPERATORS(T) \
  inline constexpr T operator|(T a, T b) noexcept { return T(uint64_t(a) | uint64_t(b)); } \
  inline T& operator|=(T& a, T b) noexcept { a = a | b; return a; } \
  inline constexpr T operator&(T a, T b) noexcept { return T(uint64_t(a) & uint64_t(b)); } \
  inline T& operator&=(T& a, T b) noexcept { a = a & b; return a; } \
  inline constexpr T operator~(T a) noexcept { return T(~uint64_t(a)); } \
  inline constexpr T operator^(T a, T b) noexcept { return T(uint64_t(a) ^ uint64_t(b)); } \
  inline T& operator^=(T& a, T b) noexcept { a = a ^ b; return a; }

// SAL annotations used by the reference's headers: all empty here.
#define _In_
#define _In_z_
#define _In_opt_
#define _In_opt_z_
#define _Out_
#define _Out_opt_
#define _Inout_
#define _Inout_opt_
#define _In_reads_(x)
#define _In_reads_opt_(x)
#define _In_reads_bytes_(x)
#define _In_reads_bytes_opt_(x)
#define _In_count_(x)
#define _In_range_(a, b)
#define _Out_writes_(x)
#define _Out_writes_opt_(x)
#define _Out_writes_all_(x)
#define _Out_writes_bytes_(x)
#define _Out_writes_bytes_opt_(x)
#define _Out_writes_bytes_to_opt_(a, b)
#define _Out_writes_to_(a, b)
#define _Out_writes_to_opt_(a, b)
#define _Inout_updates_(x)
#define _Inout_updates_all_(x)
#define _Inout_updates_all_opt_(x)
#define _Inout_updates_bytes_(x)
#define _Outptr_
#define _Outptr_opt_
#define _COM_Outptr_
#define _COM_Outptr_opt_
#define _Reserved_
#define _Success_(x)
#define _When_(a, b)
#define _Use_decl_annotations_
#define _Analysis_assume_(x)
#define _Ret_maybenull_
#define _Check_return_
