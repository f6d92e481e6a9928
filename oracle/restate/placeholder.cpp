// TEST INFRASTRUCTURE ONLY. liboracle.so anchor; the restated codecs live in the sibling files.
extern "C" int dxtex_oracle_abi_version() { return 1; }
