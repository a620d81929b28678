from .backend import (PagedAttnBackend, PagedAttnImpl, PagedAttnMetadata,  # noqa: F401
                      PagedAttnMetadataBuilder, compute_slot_mapping,
                      compute_slot_mapping_start_idx)
