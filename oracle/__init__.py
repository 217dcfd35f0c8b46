"""CPU oracle - TEST INFRASTRUCTURE ONLY (see oracle/abrk_oracle.c)."""
