"""Command-line drop-ins for the reference's dedup scripts (same flags, same artifacts)."""
