// TEST INFRASTRUCTURE ONLY (oracle/_ref). Compiles the reference's DirectXTex/DirectXTexConvert.cpp unmodified and in
// place (-I$(REF)/DirectXTex, see oracle/Makefile) against oracle/shim; no reference source is copied into this repository.
#include "DirectXTexConvert.cpp"
