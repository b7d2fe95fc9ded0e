// Minimal stand-in for the fixed-function GL entry points NeRF::DrawCPUMesh uses -- TEST INFRASTRUCTURE ONLY; tests/compat_driver.cpp defines them
// as call recorders.
#pragma once
typedef unsigned int GLenum; typedef int GLint; typedef int GLsizei; typedef void GLvoid;
#define GL_VERTEX_ARRAY 0x8074
#define GL_NORMAL_ARRAY 0x8075
#define GL_COLOR_ARRAY 0x8076
#define GL_FLOAT 0x1406
#define GL_UNSIGNED_BYTE 0x1401
#define GL_UNSIGNED_INT 0x1405
#define GL_TRIANGLES 0x0004
extern "C" {
void glEnableClientState(GLenum cap);
void glDisableClientState(GLenum cap);
void glVertexPointer(GLint size, GLenum type, GLsizei stride, const GLvoid* ptr);
void glColorPointer(GLint size, GLenum type, GLsizei stride, const GLvoid* ptr);
void glNormalPointer(GLenum type, GLsizei stride, const GLvoid* ptr);
void glDrawElements(GLenum mode, GLsizei count, GLenum type, const GLvoid* indices);
}
