"""MI355X-native particle advect + locate loop behind cudaParticlesFoam's call surface."""
__version__ = "0.1.0"
