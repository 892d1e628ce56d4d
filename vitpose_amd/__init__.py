"""vitpose_amd: MI355X-native (gfx950) hot path of ViTPose: ViT backbone +
TopdownHeatmapSimpleHead behind the reference's registry surface."""
__version__ = '0.1.0'
