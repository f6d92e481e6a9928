// TEST INFRASTRUCTURE ONLY (oracle). Intentionally empty stand-in for <directx/d3d12.h>
// (DirectXTexP.h:141). __d3d12_h__ is deliberately NOT defined so the D3D overloads stay out.
#pragma once
