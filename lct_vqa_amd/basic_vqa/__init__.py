"""Mirror of the reference's ``basic_vqa`` plugin surface (model_factory / architect_factory /
models / models_lct / config) backed by the HIP library.  Same function names, argument meaning,
module attribute names and state_dict keys as /root/reference/basic_vqa."""
