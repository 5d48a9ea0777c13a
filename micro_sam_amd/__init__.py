"""MI355X (gfx950) core behind micro_sam's hot path; mirrors the API of micro_sam 1.8.10."""
__version__ = "1.8.10"
