"""TEST INFRASTRUCTURE: CPU oracle package (see bz2_oracle.c). Not product code."""
