"""darts_vqa/pcdarts: the search network is shared with basic_vqa (the reference's two copies of
operations.py / genotypes.py are byte-identical; model_search.py differs only by the optional vqa_model
back-reference and save/load_arch_parameters, both supported by the shared implementation)."""
from ...basic_vqa.pcdarts import genotypes, model_search, operations  # noqa: F401
