// MI355XContext.h - the context the bound entry points share (reference-side binding code).
//
// The reference's GPU path owns ONE GPUCompressBC for a whole CompressEx call and calls Prepare once per mip size
// (DirectXTexCompressGPU.cpp:392-442; BCDirectCompute.cpp:203-369 allocates the per-size buffers there, :373-642 reuses them). The
// counterpart here is a dxtex_ctx: a stream plus grow-only staging and BC6H / BC7 search scratch (about 2.3 GiB for a 4096^2 BC7 image).
// Creating and destroying one per call would allocate and free all of that around every texture, so the binding keeps one context per
// (host thread, HIP device) for the lifetime of the thread: a context is single-threaded by contract (include/dxtex_amd.h), a
// thread_local cache makes the bound functions re-entrant without a lock, and dxtex_ctx_prepare - called per size as Prepare is - finds
// the buffers already there on every call after the first of a size.
#pragma once
#include <dxtex_amd.h>
#include <vector>

namespace DirectX
{
    namespace MI355X
    {
        // nullptr when `device` is not a usable gfx950 device (the bound functions return E_FAIL: there is no CPU fallback to take)
        inline dxtex_ctx* ContextFor(int device) noexcept
        {
            struct Slot { int device; dxtex_ctx* ctx; };
            struct Cache
            {
                std::vector<Slot> slots;
                ~Cache() { for (const Slot& s : slots) dxtex_ctx_destroy(s.ctx); }
            };
            thread_local Cache cache;
            for (const Slot& s : cache.slots)
                if (s.device == device) return s.ctx;
            dxtex_ctx* ctx = nullptr;
            if (dxtex_ctx_create(device, &ctx) != DXTEX_S_OK) return nullptr;
            try { cache.slots.push_back(Slot{ device, ctx }); }
            catch (...) { dxtex_ctx_destroy(ctx); return nullptr; }
            return ctx;
        }
    }
}
