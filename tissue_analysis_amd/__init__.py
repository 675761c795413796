"""tissue_analysis_amd -- MI355X-native per-label voxel scan behind the SpatialImageAnalysis API.

    from tissue_analysis_amd import SpatialImageAnalysis, DICT
    analysis = SpatialImageAnalysis(labelled_volume, ignoredlabels=0, return_type=DICT, background=1)
    analysis.volume(); analysis.center_of_mass(); analysis.neighbors(); analysis.wall_areas()
    analysis.inertia_axis()

The voxel work runs in hand-written HIP kernels for gfx950 (csrc/), reached through the C ABI in
include/tissue_scan.h; there is no CPU fallback.
"""
from .spatial_image import SpatialImage
from .spatial_image_analysis import (NPLIST, LIST, DICT, AbstractSpatialImageAnalysis,
                                     SpatialImageAnalysis3D, SpatialImageAnalysis, dilation,
                                     dilation_by, real_indices, return_list_of_vectors, hollow_out_cells, wall,
                                     contact_surface, coordinates_centering3D, compute_covariance_matrix,
                                     eigen_values_vectors, distance)
from .extraction import Extraction, extract_volume
from .property_graph import PropertyGraph
from .graph_from_image import graph_from_image, property_graph_to_dataframe

__all__ = ["SpatialImage", "NPLIST", "LIST", "DICT", "AbstractSpatialImageAnalysis",
           "SpatialImageAnalysis3D", "SpatialImageAnalysis", "Extraction", "extract_volume",
           "dilation", "dilation_by", "real_indices", "return_list_of_vectors", "hollow_out_cells", "wall", "contact_surface",
           "coordinates_centering3D", "compute_covariance_matrix", "eigen_values_vectors", "distance",
           "PropertyGraph", "graph_from_image", "property_graph_to_dataframe"]
__version__ = "0.1.0"
