"""The dense-mapping knobs of pyslam/config_parameters.py:286-380, same names and defaults.

When real pySLAM is importable its Parameters class is used instead (see get_parameters()), so a
user's edits to config_parameters.py keep working."""


class Parameters:
    kDenseMappingDtypeVertices = "float32"
    kDenseMappingDtypeColors = "float32"
    kDenseMappingDtypeDepth = "float32"
    kDenseMappingDtypeSemantics = "int32"
    kDenseMappingDtypeObjectIds = "int32"
    kDenseMappingDtypeTriangles = "uint32"

    kDepthImageUndistortionUseOptimalNewCameraMatrixWithAlphaScale = True  # config_parameters.py:282-285
    kDepthImageUndistortionOptimalNewCameraMatrixWithAlphaScaleValue = 0.7

    kDoVolumetricIntegration = False
    kVolumetricIntegrationType = "VOXEL_GRID"
    kVolumetricIntegrationVoxelLength = 0.015  # [m]
    kVolumetricIntegrationUseVoxelBlocks = True
    kVolumetricIntegrationBlockSize = 8
    kVolumetricIntegrationTBBThreads = 2  # no meaning on the GPU path; kept for API parity
    kVolumetricIntegrationFpsThrottleEnabled = True
    kVolumetricIntegrationFpsThrottleMinQueueSize = 10
    kVolumetricIntegrationFpsMaxThreshold = 10.0
    kVolumetricIntegrationFpsThrottleBaseDelay = 0.01
    kVolumetricIntegrationFpsThrottleScale = 0.1
    kVolumetricIntegrationVoxelGridMinCount = 3
    kVolumetricIntegrationVoxelGridMinConfidence = 0.6
    kVolumetricIntegrationVoxelGridUseCarving = False
    kVolumetricIntegrationVoxelGridCarvingDepthMin = 1e-2
    kVolumetricIntegrationVoxelGridCarvingDepthMaxIndoor = 8.0
    kVolumetricIntegrationVoxelGridCarvingDepthMaxOutdoor = 15.0
    kVolumetricIntegrationVoxelGridCarvingDepthThreshold = 3e-2
    kVolumetricIntegrationVoxelGridShadowPointsFilter = True
    kVolumetricIntegrationTsdfExtractMesh = True
    kVolumetricIntegrationTSdfTrunc = 0.04
    kVolumetricIntegrationTsdfDepthTruncIndoor = 4.0
    kVolumetricIntegrationTsdfDepthTruncOutdoor = 10.0
    kVolumetricIntegrationMinNumLBATimes = 1
    kVolumetricIntegrationOutputTimeInterval = 1.0
    kVolumetricIntegrationUseDepthEstimator = False
    kVolumetricIntegrationDepthEstimatorType = "DEPTH_RAFT_STEREO"
    kVolumetricIntegrationDepthEstimationFilterShadowPoints = True
    # semantic integration, config_parameters.py:364-380
    kVolumetricSemanticProbabilisticIntegrationUseDepth = True
    kVolumetricSemanticProbabilisticIntegrationDepthThresholdIndoor = 5.0
    kVolumetricSemanticProbabilisticIntegrationDepthThresholdOutdoor = 10.0
    kVolumetricSemanticProbabilisticIntegrationDepthDecayRateIndoor = 0.1
    kVolumetricSemanticProbabilisticIntegrationDepthDecayRateOutdoor = 0.05
    kVolumetricSemanticIntegrationUseInstanceIds = True
    kVolumetricSemanticIntegrationMinVoteRatio = 0.5
    kVolumetricSemanticIntegrationMinVotes = 3
    kDoSparseSemanticMappingAndSegmentation = False
    kMultiprocessingProcessJoinDefaultTimeout = 5.0
    kLoopDetectingTimeoutPopKeyframe = 0.5

    # GPU-path additions (no reference counterpart)
    kVolumetricIntegrationHipDevice = 0
    kVolumetricIntegrationHipMaxBlocks = None  # None: library default pool size
    kVolumetricIntegrationUseSharedMemory = True  # keyframe images / output arrays through shared memory, queues carry control only
    kVolumetricIntegrationSharedMemorySlots = 96  # ring slots (one keyframe each; 5.5 MB at 640x480)
    # TSDF output ticks: False = Open3D's float64 vertices / colours, as the reference hands them on (base.py:209-214); True = extract
    # them in kDenseMappingDtypeVertices / kDenseMappingDtypeColors (float32: what the viewer casts them to) - the float64 values
    # rounded once on the device (hv_tsdf_extract_mesh_f32), a third fewer bytes per tick.  Saved files always come from float64.
    kVolumetricIntegrationTsdfOutputInDenseMappingDtype = False


def get_parameters():
    try:  # real pySLAM present: honour its (possibly edited) constants, add ours
        from pyslam.config_parameters import Parameters as P  # type: ignore

        for k, v in vars(Parameters).items():
            if k.startswith("k") and not hasattr(P, k):
                setattr(P, k, v)
        return P
    except Exception:
        return Parameters


def static_fields_to_dict(cls):
    return {k: v for k, v in vars(cls).items() if k.startswith("k")}
