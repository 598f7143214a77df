"""Drop-in replacements for the reference's codes/models/modules hot-path files:
same module names, class names, constructor / forward signatures and
state_dict keys; forward runs on the HIP kernels of libselfc_hip.so."""
