// TEST INFRASTRUCTURE ONLY (oracle). Stand-in for <wsl/wrladapter.h> (DirectXTexP.h:140).
#pragma once
namespace Microsoft { namespace WRL { template<class T> class ComPtr; } }
