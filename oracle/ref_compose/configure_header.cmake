# cmake -DAMRVOLUMERENDERER_APP_NAME=<name> -DTEMPLATE=<.../amrVolumeRendererConfig.h.in> -DOUTPUT=<file> -P configure_header.cmake
# The one header the reference generates at configure time, made by the command its own build
# uses for it (configure_file, /root/reference CMake/amrVolumeRendererMacros.cmake) on its own
# template -- nothing of the reference's build system is run.
if(NOT DEFINED TEMPLATE OR NOT DEFINED OUTPUT OR NOT DEFINED AMRVOLUMERENDERER_APP_NAME)
  message(FATAL_ERROR "TEMPLATE, OUTPUT and AMRVOLUMERENDERER_APP_NAME must be given")
endif()
configure_file("${TEMPLATE}" "${OUTPUT}")
