"""map_neo_amd -- MI355X-native MinHash-LSH near-dedup (the document near-deduplication
stage of MAP-NEO's Matrix pipeline).  Python host -> ctypes -> libmdd_hip.so (hand-written
HIP for gfx950).  There is no CPU fallback: importing the package is cheap, but any compute
call raises if the HIP library or a GPU is missing."""
__all__ = ["lib", "minhash", "dedup", "corpus"]
__version__ = "0.1.0"
