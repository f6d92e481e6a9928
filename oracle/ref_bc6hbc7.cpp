// TEST INFRASTRUCTURE ONLY (oracle/_ref). Compiles the reference's DirectXTex/BC6HBC7.cpp unmodified and in
// place: the quoted include resolves through -I$(REF)/DirectXTex (see oracle/Makefile); no reference
// source is copied into this repository. Angle includes inside it resolve to oracle/shim/.
#include "BC6HBC7.cpp"
