"""PC-DARTS search network on the HIP library: mirror of reference basic_vqa/pcdarts/ (operations,
genotypes, model_search, architect, architect_lct) with the same class names, constructor signatures,
attribute names and state_dict keys."""
