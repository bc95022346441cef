from smirk_amd.masking import (load_probabilities_per_FLAME_triangle, triangle_area, random_barycentric, masking, point2ind,  # noqa: F401
                               transfer_pixels, mesh_based_mask_uniform_faces)
