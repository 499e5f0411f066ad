"""tutel_amd -- MI355X (gfx950) native implementation of Tutel's MoE forward hot path behind
Tutel's own Python API.  `import tutel` (the alias package at the repo root) resolves
`tutel.moe`, `tutel.net`, `tutel.system`, `tutel.impls.*`, ... to the modules in here."""
__version__ = "0.1.0"
