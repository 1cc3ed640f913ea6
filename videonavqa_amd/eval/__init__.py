"""Counterparts of the reference's `eval/` package for the MI355X path (same CLI flags, constants,
dataset contract and checkpoint schema)."""
