"""TEST INFRASTRUCTURE ONLY: CPU oracle for the TERSE/PROLIX hot path (see terse_oracle.c)."""
