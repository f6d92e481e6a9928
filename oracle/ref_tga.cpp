// TEST INFRASTRUCTURE ONLY (oracle/_ref). Compiles the reference's DirectXTex/DirectXTexTGA.cpp unmodified and in
// place (-I$(REF)/DirectXTex, see oracle/Makefile); no reference source is copied into this repository.
#include "DirectXTexTGA.cpp"
