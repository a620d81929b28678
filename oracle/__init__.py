"""CPU checkers for the HIP path -- test infrastructure only (see oracle/README.md)."""
