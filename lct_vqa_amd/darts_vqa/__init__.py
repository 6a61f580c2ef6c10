"""Mirror of the reference's ``darts_vqa`` plugin surface (factory / vqa_model / pcdarts.architect_vqa / config)
backed by the HIP library: same names, signatures, attribute names and state_dict keys as /root/reference/darts_vqa."""
